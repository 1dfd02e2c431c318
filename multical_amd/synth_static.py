"""Synthetic static-camera calibration problems: BASELINE.json configs[0] (kalibr_calibrate_cameras, one pinhole-radtan
camera, ~500 views of one 6x6 Aprilgrid) and configs[2] (4 cameras, 6 boards, joint intrinsics + extrinsics).

Problem shape = kalibr_camera_calibration/CameraCalibrator.py:195-219 (`addTargetView`): per view one target pose
design-variable pair T_t_c (RotationQuaternion + EuclideanPoint), per camera a projection and a distortion design
variable (`CameraDesignVariable`, all active), per extra camera one baseline transformation; every corner gives a
ReprojectionError over  B_{j-1} * ... * B_0 * inverse(T_t_c) * p  with a Blake-Zisserman(2) M-estimator.
With several boards every (view, board) pair has its own target pose.  Measurements come from the forward model itself
(`synthesize`: evaluate at the ground truth with zero measurements), as in synth.py."""
import numpy as np

from . import spec as S
from .synth import (CAM_INTRINSICS, C_CAM_BODY, _perturb_q, _small_rotation, aprilgrid_points, quat_from_matrix)


class StaticConfig(object):
    def __init__(self, **kw):
        self.seed = 20221023
        self.n_cams = 1
        self.n_boards = 1
        self.n_views = 500
        self.cam_models = None
        self.corner_stride = 1
        self.pixel_noise = 0.3
        self.blake_zisserman = 2
        self.perturb = 1.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown StaticConfig field %r" % k)
            setattr(self, k, v)


def baseline_static_config(index, **over):
    if index == 0:
        c = dict(n_cams=1, n_boards=1, n_views=500)
    elif index == 2:
        c = dict(n_cams=4, n_boards=6, n_views=500)      # 4 x 500 = ~2k frames
    else:
        raise ValueError("static problems are BASELINE configs[0] and configs[2]")
    c["seed"] = 20221023 + index + 1
    c.update(over)
    return StaticConfig(**c)


def make_static_world(cfg):
    rng = np.random.Generator(np.random.PCG64(cfg.seed))
    pts = aprilgrid_points()
    ext = pts.max(0)
    W = dict(cfg=cfg, board_points=pts)
    boards = []
    for b in range(cfg.n_boards):
        az = np.deg2rad(-40 + 80.0 * (b + 0.5) / cfg.n_boards)
        el = np.deg2rad(10.0 * ((b % 3) - 1)) if cfg.n_boards > 1 else 0.0
        d = 1.6 + 0.2 * (b % 2)
        center = d * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        zax = -center / np.linalg.norm(center)
        xax = np.cross([0.0, 0.0, 1.0], zax); xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        R_w_p = np.stack([xax, yax, zax], 1) @ _small_rotation(rng, 0.05)
        t_w_p = center - R_w_p @ (0.5 * ext)
        boards.append(dict(C=R_w_p.T, t=-R_w_p.T @ t_w_p))       # T_p_w
    W["boards"] = boards
    models = cfg.cam_models or [(S.PROJ_PINHOLE, S.DIST_RADTAN)]
    cams, T_c_c0 = [], [(np.eye(3), np.zeros(3))]
    for j in range(cfg.n_cams):
        model, dist = models[j % len(models)]
        proj, dk = CAM_INTRINSICS[(model, dist)]
        cams.append(dict(model=model, dist=dist, proj=np.array(proj), dk=None if dk is None else np.array(dk)))
        if j > 0:
            Cb = _small_rotation(rng, 0.08)
            tb = np.array([-0.11, 0.0, 0.0]) + rng.uniform(-0.01, 0.01, 3)
            cams[j]["baseline"] = (quat_from_matrix(Cb), tb, Cb)      # T_cj_c(j-1)
            Cp, tp = T_c_c0[-1]
            T_c_c0.append((Cb @ Cp, Cb @ tp + tb))
    W["cams"] = cams
    # rig poses: camera 0 looks along world +x, jittered
    views = []
    for v in range(cfg.n_views):
        R_w_c0 = _small_rotation(rng, 0.25) @ C_CAM_BODY.T
        t_w_c0 = rng.uniform([-0.4, -0.5, -0.3], [0.4, 0.5, 0.3])
        views.append((R_w_c0, t_w_c0))
    W["views"] = views
    # observations: per camera a list of (view, board, corner indices); target-pose dvs per (view, board) seen by any camera
    fu, fv, cu, cv = 460.0, 460.0, 376.0, 240.0
    seen, obs = {}, [[] for _ in cams]
    for v, (R_w_c0, t_w_c0) in enumerate(views):
        for b, brd in enumerate(boards):
            pw = (brd["C"].T @ (pts - brd["t"]).T).T                 # board points in the world
            pc0 = (pw - t_w_c0) @ R_w_c0                              # R_w_c0^T (p - t)
            for j in range(cfg.n_cams):
                Cj, tj = T_c_c0[j]
                pc = pc0 @ Cj.T + tj
                z = pc[:, 2]
                u = fu * pc[:, 0] / np.maximum(z, 1e-3) + cu
                w_ = fv * pc[:, 1] / np.maximum(z, 1e-3) + cv
                vis = (z > 0.3) & (u > 40) & (u < 712) & (w_ > 40) & (w_ < 440)
                vis[np.arange(pts.shape[0]) % cfg.corner_stride != 0] = False
                if vis.sum() >= 4:
                    obs[j].append((v, b, np.nonzero(vis)[0]))
                    seen.setdefault((v, b), len(seen))
    W["obs"], W["target_index"] = obs, seen
    # ground-truth target poses T_t_c0 = T_p_w * T_w_c0
    W["targets"] = {}
    for (v, b), k in seen.items():
        R_w_c0, t_w_c0 = views[v]
        C = boards[b]["C"] @ R_w_c0
        t = boards[b]["C"] @ t_w_c0 + boards[b]["t"]
        W["targets"][k] = (quat_from_matrix(C), t)
    return W


def static_truth(W):
    return dict(cams=[dict(proj=c["proj"].copy(), dk=None if c["dk"] is None else c["dk"].copy(),
                           baseline=None if "baseline" not in c else (c["baseline"][0].copy(), c["baseline"][1].copy())) for c in W["cams"]],
                targets={k: (q.copy(), t.copy()) for k, (q, t) in W["targets"].items()})


def static_initial(W):
    cfg = W["cfg"]
    rng = np.random.Generator(np.random.PCG64(cfg.seed + 77))
    s = 0.01 * cfg.perturb
    st = static_truth(W)
    for c in st["cams"]:
        c["proj"] = c["proj"] * (1.0 + rng.normal(0, 0.5 * s, c["proj"].size))
        if c["dk"] is not None:
            c["dk"] = c["dk"] + rng.normal(0, 0.2 * s, 4)
        if c["baseline"] is not None:
            c["baseline"] = (_perturb_q(c["baseline"][0], rng, s), c["baseline"][1] + rng.normal(0, s, 3))
    for k in st["targets"]:
        q, t = st["targets"][k]
        st["targets"][k] = (_perturb_q(q, rng, s), t + rng.normal(0, s, 3))
    return st


def build_static_spec(W, state, meas, with_mest=True):
    cfg = W["cfg"]
    sp = S.ProblemSpec()
    H = dict(cams=[], targets={})
    for j, cam in enumerate(W["cams"]):
        sc = state["cams"][j]
        hp = sp.add_dv(S.DV_PROJECTION, sc["proj"], True, 1.0, S.CALIBRATION_GROUP_ID)
        hk = sp.add_dv(S.DV_DISTORTION, sc["dk"], True, 1.0, S.CALIBRATION_GROUP_ID) if sc["dk"] is not None else -1
        hc = sp.add_camera(cam["model"], cam["dist"], hp, hk)
        hq = ht = -1
        if sc["baseline"] is not None:
            hq = sp.add_dv(S.DV_ROTATION_QUATERNION, sc["baseline"][0], True, 1.0, S.CALIBRATION_GROUP_ID)
            ht = sp.add_dv(S.DV_EUCLIDEAN_POINT, sc["baseline"][1], True, 1.0, S.CALIBRATION_GROUP_ID)
        H["cams"].append(dict(proj=hp, dist=hk, camera=hc, bq=hq, bt=ht))
    for k in range(len(W["targets"])):
        q, t = state["targets"][k]
        H["targets"][k] = (sp.add_dv(S.DV_ROTATION_QUATERNION, q, True, 1.0, S.HELPER_GROUP_ID),
                           sp.add_dv(S.DV_EUCLIDEAN_POINT, t, True, 1.0, S.HELPER_GROUP_ID))
    pts = W["board_points"]
    mest, mp = S.MEST_NONE, 0.0
    if with_mest and cfg.blake_zisserman > 0:
        mest, mp = S.MEST_BLAKE_ZISSERMAN, S.blake_zisserman_epsilon(int(cfg.blake_zisserman))
    for j, cam in enumerate(W["cams"]):
        ob = W["obs"][j]
        if not ob:
            continue
        first = np.concatenate([[0], np.cumsum([idx.size for _, _, idx in ob])]).astype(np.int32)
        p = np.concatenate([pts[idx] for _, _, idx in ob], 0)
        tq = [H["targets"][W["target_index"][(v, b)]][0] for v, b, _ in ob]
        tt = [H["targets"][W["target_index"][(v, b)]][1] for v, b, _ in ob]
        chain_q = [H["cams"][i]["bq"] for i in range(1, j + 1)]
        chain_t = [H["cams"][i]["bt"] for i in range(1, j + 1)]
        sp.add_static_reprojection_batch(H["cams"][j]["camera"], chain_q, chain_t, tq, tt, first, meas[j], p, np.ones(first[-1]), mest, mp)
    return sp, H


def synthesize_static(W, make_api, noise=True):
    cfg = W["cfg"]
    zero = [np.zeros((sum(idx.size for _, _, idx in ob), 2)) for ob in W["obs"]]
    spec, _ = build_static_spec(W, static_truth(W), zero, with_mest=False)
    api = make_api()
    spec.emit(api)
    api.finalize()
    _, e = api.evaluate_error(False, True)
    rng = np.random.Generator(np.random.PCG64(cfg.seed + 1234))
    meas, row = [], 0
    for j, ob in enumerate(W["obs"]):
        n = zero[j].shape[0]
        m = e[row:row + 2 * n].reshape(n, 2).copy()       # e_out = -(0 - yhat)
        row += 2 * n
        if noise:
            m += rng.normal(0, cfg.pixel_noise, m.shape)
        meas.append(m)
    assert row == e.size
    return meas


def make_static_problem(cfg, make_api, noise=True):
    W = make_static_world(cfg)
    meas = synthesize_static(W, make_api, noise)
    spec, H = build_static_spec(W, static_initial(W), meas)
    return spec, H, W, meas


def with_landmarks(spec, seed=2, inactive_every=7, constant_every=5, flip_w_every=0):
    """a copy of a static spec whose target corners are HomogeneousPoint design variables shared by all cameras and views
    (every `inactive_every`-th landmark inactive, every `constant_every`-th corner left a constant)"""
    import copy
    from . import spec as _S
    rng = np.random.RandomState(seed)
    sp = copy.deepcopy(spec)
    pts = {}
    for i, k, d in list(sp.batch_ops()):
        if k != "static_reproj":
            continue
        handles = []
        for row in d["p"]:
            key = tuple(np.round(row, 9))
            if key not in pts:
                n = len(pts)
                if constant_every and n % constant_every == constant_every - 1:
                    pts[key] = None
                else:
                    ph = np.array([row[0], row[1], row[2], 1.0]) + np.r_[0.002 * rng.standard_normal(3), 0.0]
                    if flip_w_every and n % flip_w_every == 0:
                        ph = -ph                       # same point, w < 0
                    pts[key] = sp.add_dv(_S.DV_HOMOGENEOUS_POINT, ph, not (inactive_every and n % inactive_every == 0), 1.0,
                                         _S.HELPER_GROUP_ID)
            handles.append(pts[key])
        sp.set_landmarks(i, handles)
    return sp
