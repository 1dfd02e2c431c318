"""Synthetic Multical problems in the shape the Python front-end builds them.

`make_world` lays out the ground truth (pose spline, boards, sensor extrinsics, time offsets, biases) and the
measurement structure (which corner is seen in which image, IMU / LiDAR sample times); `build_spec` turns a world
plus a state (truth or perturbed initial guess) plus measurements into a `ProblemSpec` in the reference's
design-variable and error-term order (calibrator.py:94-123, 214-231).  Measurements are synthesised by running
the forward model itself on the ground-truth state with zero measurements (`synthesize`), so no third
implementation of the sensor models exists: on the GPU box the forward model is the HIP backend, in the CPU tests
it is the oracle.
"""
import numpy as np

from . import spec as S

SPLINE_ORDER = 6


class SynthConfig(object):
    def __init__(self, **kw):
        self.seed = 20221023
        self.duration = 10.0
        self.pose_knots_per_second = 100
        self.bias_knots_per_second = 5
        self.time_padding = 0.03
        self.n_cams = 1
        self.cam_rate = 20.0
        self.cam_models = None            # list of (projection, distortion); default pinhole-radtan
        self.n_boards = 6
        self.corner_stride = 1            # keep every n-th corner (tests use > 1)
        self.max_corners_per_obs = 144
        self.n_imus = 1
        self.imu_rate = 200.0
        self.n_lidars = 0
        self.lidar_rate = 10.0
        self.lidar_points_per_frame = 300
        self.time_calibration = True
        self.bias_motion_terms = True
        self.pose_motion_terms = False
        self.huber_accel = -1.0
        self.huber_gyro = -1.0
        self.blake_zisserman_cam = -1.0
        self.lidar_mest = None
        self.reproj_sigma = 1.0
        self.pixel_noise = 0.3
        self.accel_sigma = 0.28
        self.gyro_sigma = 0.024
        self.lidar_sigma = 0.02
        self.motion_speed = 1.0           # time scale of the trajectory (tests use > 1 for well-excited lever arms)
        self.perturb = 1.0                # scale of the initial-guess perturbation
        self.t_start = 0.0                # window start (multi-GPU weak scaling shifts nothing: one long trajectory)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown SynthConfig field %r" % k)
            setattr(self, k, v)


# the five BASELINE.json configurations (config 1 and 3 are the static-camera problems, see make_static_world)
def baseline_config(index, **over):
    if index == 1:
        c = dict(duration=60.0, n_cams=1, n_imus=1, n_lidars=0)
    elif index == 3:
        c = dict(duration=120.0, n_cams=4, n_imus=3, n_lidars=2)
    elif index == 4:
        c = dict(duration=600.0, n_cams=16, n_imus=8, n_lidars=4, imu_rate=200.0)
    else:
        raise ValueError("baseline_config handles the spline problems: configs[1], [3], [4]")
    c["seed"] = 20221023 + index + 1
    c.update(over)
    return SynthConfig(**c)


def trajectory(t, speed=1.0):
    t = np.asarray(t, dtype=np.float64) * speed
    r = np.stack([1.5 * np.sin(0.4 * t), 1.0 * np.sin(0.31 * t + 0.7), 0.3 * np.sin(0.53 * t)], 0)
    th = np.stack([0.3 * np.sin(0.37 * t), 0.25 * np.sin(0.29 * t + 1.1), 0.6 * np.sin(0.21 * t)], 0)
    return np.concatenate([r, th], 0)   # 6 x n


def rotvec_to_matrix(th):
    """Batched Rodrigues; th: (..., 3) -> (..., 3, 3).  Only used to decide what is visible."""
    th = np.asarray(th, dtype=np.float64)
    ang = np.linalg.norm(th, axis=-1, keepdims=True)
    ax = th / np.maximum(ang, 1e-300)
    x, y, z = ax[..., 0], ax[..., 1], ax[..., 2]
    K = np.zeros(th.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0], K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -z, y, z, -x, -y, x
    s, c = np.sin(ang)[..., None], np.cos(ang)[..., None]
    # matches RotationVector::parametersToRotationMatrix, which is the transpose of the usual exp(theta^)
    R = np.eye(3) + s * K + (1 - c) * (K @ K)
    return np.swapaxes(R, -1, -2)


def quat_from_matrix(C):
    """r2quat (sm_kinematics/src/quaternion_algebra.cpp:16-75) -- [x y z w] with quat2r(q) == C."""
    c1, c2, c3, c4, c5, c6, c7, c8, c9 = C[0, 0], C[1, 0], C[2, 0], C[0, 1], C[1, 1], C[2, 1], C[0, 2], C[1, 2], C[2, 2]
    dc = np.abs([1.0 + c1 - c5 - c9, 1.0 - c1 + c5 - c9, 1.0 - c1 - c5 + c9, 1.0 + c1 + c5 + c9])
    m = int(np.argmax(dc))
    q = np.zeros(4)
    if m == 0:
        q[0] = 0.5 * np.sqrt(dc[0]); c = 0.25 / q[0]; q[1] = c * (c4 + c2); q[2] = c * (c7 + c3); q[3] = c * (c8 - c6)
    elif m == 1:
        q[1] = 0.5 * np.sqrt(dc[1]); c = 0.25 / q[1]; q[0] = c * (c4 + c2); q[2] = c * (c6 + c8); q[3] = c * (c3 - c7)
    elif m == 2:
        q[2] = 0.5 * np.sqrt(dc[2]); c = 0.25 / q[2]; q[0] = c * (c3 + c7); q[1] = c * (c6 + c8); q[3] = c * (c4 - c2)
    else:
        q[3] = 0.5 * np.sqrt(dc[3]); c = 0.25 / q[3]; q[0] = c * (c8 - c6); q[1] = c * (c3 - c7); q[2] = c * (c4 - c2)
    if q[3] < 0:
        q = -q
    return q


def _small_rotation(rng, scale):
    return rotvec_to_matrix(rng.uniform(-scale, scale, 3))


def aprilgrid_points(rows=6, cols=6, size=0.088, spacing=0.3):
    pts = []
    step = size * (1.0 + spacing)
    for r in range(rows):
        for c in range(cols):
            x0, y0 = c * step, r * step
            pts += [[x0, y0, 0.0], [x0 + size, y0, 0.0], [x0 + size, y0 + size, 0.0], [x0, y0 + size, 0.0]]
    return np.array(pts)


CAM_INTRINSICS = {
    (S.PROJ_PINHOLE, S.DIST_RADTAN): ([460.0, 460.0, 376.0, 240.0], [-0.28, 0.07, 2e-4, 2e-5]),
    (S.PROJ_PINHOLE, S.DIST_EQUI): ([460.0, 460.0, 376.0, 240.0], [-0.01, 0.02, -0.01, 0.002]),
    (S.PROJ_PINHOLE, S.DIST_NONE): ([460.0, 460.0, 376.0, 240.0], None),
    (S.PROJ_OMNI, S.DIST_RADTAN): ([1.2, 900.0, 900.0, 376.0, 240.0], [-0.05, 0.01, 2e-4, 2e-5]),
    (S.PROJ_OMNI, S.DIST_NONE): ([1.2, 900.0, 900.0, 376.0, 240.0], None),
    (S.PROJ_OMNI, S.DIST_EQUI): ([1.2, 900.0, 900.0, 376.0, 240.0], [-0.01, 0.02, -0.01, 0.002]),
    (S.PROJ_DS, S.DIST_NONE): ([-0.2, 0.6, 350.0, 350.0, 376.0, 240.0], None),
}
C_CAM_BODY = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])   # camera z = body x


def make_world(cfg):
    rng = np.random.Generator(np.random.PCG64(cfg.seed))
    W = dict(cfg=cfg)
    pad = cfg.time_padding
    # ---- pose spline (ground truth IS the spline): uniform knots, 2*pad margin (sensors_and_targets.py:512-539)
    t0, t1 = cfg.t_start - 2 * pad, cfg.t_start + cfg.duration + 2 * pad
    n_seg = int(round((t1 - t0) * cfg.pose_knots_per_second))
    dt = (t1 - t0) / n_seg
    knots = t0 + (np.arange(n_seg + 2 * SPLINE_ORDER - 1) - (SPLINE_ORDER - 1)) * dt
    n_coef = n_seg + SPLINE_ORDER - 1
    grev = np.array([knots[i + 1:i + SPLINE_ORDER].mean() for i in range(n_coef)])
    W["pose_knots"], W["pose_coeffs"] = knots, trajectory(grev, cfg.motion_speed)
    W["t_min"], W["t_max"] = knots[SPLINE_ORDER - 1], knots[-SPLINE_ORDER]
    # ---- boards around +x, facing the origin
    pts = aprilgrid_points()
    W["board_points"] = pts
    ext = pts.max(0)
    boards = []
    for b in range(cfg.n_boards):
        az = np.deg2rad(-50 + 100.0 * (b + 0.5) / cfg.n_boards) + rng.uniform(-0.05, 0.05)
        el = np.deg2rad(12.0 * ((b % 3) - 1)) + rng.uniform(-0.03, 0.03)
        d = 3.0 + rng.uniform(-0.3, 0.3)
        center = d * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        zax = -center / np.linalg.norm(center)                 # board normal towards the origin
        xax = np.cross([0.0, 0.0, 1.0], zax); xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        R_w_p = np.stack([xax, yax, zax], 1) @ _small_rotation(rng, 0.1)
        t_w_p = center - R_w_p @ (0.5 * ext)
        C_p_w, t_p_w = R_w_p.T, -R_w_p.T @ t_w_p
        boards.append(dict(q=quat_from_matrix(C_p_w), t=t_p_w, C=C_p_w))
    W["boards"] = boards
    W["gravity"] = np.array([0.0, 0.0, -9.80655]) if True else None
    # ---- cameras
    models = cfg.cam_models or [(S.PROJ_PINHOLE, S.DIST_RADTAN)]
    cams = []
    for c in range(cfg.n_cams):
        model, dist = models[c % len(models)]
        proj, dk = CAM_INTRINSICS[(model, dist)]
        C_c_b = C_CAM_BODY @ _small_rotation(rng, 0.3 if c else 0.05)
        t_c_b = rng.uniform(-0.2, 0.2, 3)
        cams.append(dict(model=model, dist=dist, proj=np.array(proj), dk=None if dk is None else np.array(dk),
                         q=quat_from_matrix(C_c_b), t=t_c_b, C=C_c_b, toff=rng.uniform(-5e-3, 5e-3) if cfg.time_calibration else 0.0))
    W["cams"] = cams
    # ---- visibility (approximate pose from the analytic trajectory, ideal pinhole + margin)
    n_frames = int(np.floor(cfg.duration * cfg.cam_rate))
    for c, cam in enumerate(cams):
        ft = cfg.t_start + (np.arange(n_frames) + 0.5 * (c % 2)) / cfg.cam_rate + 0.01
        ft = ft[(ft > W["t_min"] + pad) & (ft < W["t_max"] - pad)]
        pose = trajectory(ft, cfg.motion_speed)
        C_w_b = rotvec_to_matrix(pose[3:].T)                    # n x 3 x 3
        obs_time, obs_board, obs_first, corner_idx = [], [], [0], []
        fu, fv, cu, cv = 460.0, 460.0, 376.0, 240.0
        for b, brd in enumerate(boards):
            R_w_p, t_w_p = brd["C"].T, -brd["C"].T @ brd["t"]
            pw = (R_w_p @ pts.T).T + t_w_p                       # 144 x 3
            pb = np.einsum("nji,nkj->nki", C_w_b, pw[None, :, :] - pose[:3].T[:, None, :])   # C_w_b^T (p - r)
            pc = pb @ cam["C"].T + cam["t"]
            z = pc[..., 2]
            u = fu * pc[..., 0] / np.maximum(z, 1e-3) + cu
            v = fv * pc[..., 1] / np.maximum(z, 1e-3) + cv
            vis = (z > 0.5) & (u > 40) & (u < 752 - 40) & (v > 40) & (v < 480 - 40)
            vis[:, np.arange(pts.shape[0]) % cfg.corner_stride != 0] = False
            for f in np.nonzero(vis.sum(1) >= 4)[0]:
                idx = np.nonzero(vis[f])[0][:cfg.max_corners_per_obs]
                obs_time.append(ft[f]); obs_board.append(b); corner_idx.append(idx); obs_first.append(obs_first[-1] + idx.size)
        order = np.argsort(np.array(obs_time), kind="stable") if obs_time else np.zeros(0, int)
        cam["obs_time"] = np.array(obs_time)[order] if obs_time else np.zeros(0)
        cam["obs_board"] = [obs_board[i] for i in order]
        cam["corner_idx"] = [corner_idx[i] for i in order]
        cam["obs_first"] = np.concatenate([[0], np.cumsum([corner_idx[i].size for i in order])]).astype(np.int32)
    # ---- IMUs
    imus = []
    tb0, tb1 = W["t_min"], W["t_max"]
    nb_seg = max(1, int(round((tb1 - tb0) * cfg.bias_knots_per_second)))
    bdt = (tb1 - tb0) / nb_seg
    bknots = tb0 + (np.arange(nb_seg + 2 * SPLINE_ORDER - 1) - (SPLINE_ORDER - 1)) * bdt
    nb_coef = nb_seg + SPLINE_ORDER - 1
    for i in range(cfg.n_imus):
        # the IMU stream spans the whole spline (recordings start before / end after the camera data; terms are kept
        # while t_min < t < t_max, sensors_and_targets.py:1154), so the padding coefficients are observed too
        n_imu = int(np.ceil((tb1 - tb0) * cfg.imu_rate)) + 1
        t = tb0 + (np.arange(n_imu) + 0.37) / cfg.imu_rate + 1e-4 * i
        t = t[(t > tb0) & (t < tb1)]
        C_i_b = _small_rotation(rng, 0.3) if i else np.eye(3)
        imus.append(dict(t=t, q=quat_from_matrix(C_i_b), r=rng.uniform(-0.2, 0.2, 3) if i else np.zeros(3), reference=(i == 0),
                         bias_knots=bknots,
                         gyro_bias=0.01 * rng.standard_normal((3, 1)) + np.cumsum(2e-4 * rng.standard_normal((3, nb_coef)), 1),
                         accel_bias=0.05 * rng.standard_normal((3, 1)) + np.cumsum(1e-3 * rng.standard_normal((3, nb_coef)), 1)))
    W["imus"] = imus
    # ---- LiDARs: points on the board planes, direction from the approximate pose
    lidars = []
    for l in range(cfg.n_lidars):
        C_l_b = _small_rotation(rng, 0.3)
        t_l_b = rng.uniform(-0.2, 0.2, 3)
        n_fr = int(np.floor(cfg.duration * cfg.lidar_rate))
        P = cfg.lidar_points_per_frame
        tt = (cfg.t_start + (np.arange(n_fr)[:, None] + np.sort(rng.uniform(0, 1, (n_fr, P)), 1)) / cfg.lidar_rate).ravel()
        keep = (tt > W["t_min"] + pad) & (tt < W["t_max"] - pad)
        tt = tt[keep]
        board = rng.integers(0, cfg.n_boards, tt.size)
        pb = np.zeros((tt.size, 3))
        pb[:, :2] = rng.uniform(0.05, 0.95, (tt.size, 2)) * ext[:2]
        Rwp = np.stack([b["C"].T for b in boards])[board]
        twp = np.stack([-b["C"].T @ b["t"] for b in boards])[board]
        pw = np.einsum("nij,nj->ni", Rwp, pb) + twp
        pose = trajectory(tt, cfg.motion_speed)
        C_w_b = rotvec_to_matrix(pose[3:].T)
        pbody = np.einsum("nji,nj->ni", C_w_b, pw - pose[:3].T)
        pl = pbody @ C_l_b.T + t_l_b
        direction = pl / np.linalg.norm(pl, axis=1, keepdims=True)
        lidars.append(dict(t=tt, dir=direction, board=board, q=quat_from_matrix(C_l_b), t_l_b=t_l_b,
                           toff=rng.uniform(-5e-3, 5e-3) if cfg.time_calibration else 0.0))
    W["lidars"] = lidars
    W["rng_state"] = rng.bit_generator.state
    return W


def truth_state(W):
    return dict(pose_coeffs=W["pose_coeffs"].copy(), gravity=W["gravity"].copy(),
                boards=[(b["q"].copy(), b["t"].copy()) for b in W["boards"]],
                cams=[dict(q=c["q"].copy(), t=c["t"].copy(), toff=c["toff"], proj=c["proj"].copy(),
                           dk=None if c["dk"] is None else c["dk"].copy()) for c in W["cams"]],
                imus=[dict(q=i["q"].copy(), r=i["r"].copy(), gyro_bias=i["gyro_bias"].copy(), accel_bias=i["accel_bias"].copy())
                      for i in W["imus"]],
                lidars=[dict(q=l["q"].copy(), t=l["t_l_b"].copy(), toff=l["toff"]) for l in W["lidars"]])


def _perturb_q(q, rng, s):
    from scipy.spatial.transform import Rotation as R
    dq = R.from_rotvec(rng.normal(0, s, 3)).as_quat()      # [x y z w]
    out = (R.from_quat(dq) * R.from_quat(q)).as_quat()
    return out if out[3] >= 0 else -out


def initial_state(W):
    """Truth perturbed by N(0, 0.01 m / 0.01 rad); time offsets and biases start at zero like the
    reference front-end (aopt.Scalar(0.0), initConstantSpline)."""
    cfg = W["cfg"]
    rng = np.random.Generator(np.random.PCG64(cfg.seed + 77))
    s = 0.01 * cfg.perturb
    st = truth_state(W)
    st["pose_coeffs"] += rng.normal(0, 0.2 * s, st["pose_coeffs"].shape)
    g = st["gravity"] + rng.normal(0, 5 * s, 3)
    st["gravity"] = g / np.linalg.norm(g) * np.linalg.norm(W["gravity"])
    st["boards"] = [st["boards"][0]] + [(_perturb_q(q, rng, s), t + rng.normal(0, s, 3)) for q, t in st["boards"][1:]]
    for c in st["cams"]:
        c["q"], c["t"], c["toff"] = _perturb_q(c["q"], rng, s), c["t"] + rng.normal(0, s, 3), 0.0
    for k, i in enumerate(st["imus"]):
        if not W["imus"][k]["reference"]:
            i["q"], i["r"] = _perturb_q(i["q"], rng, s), i["r"] + rng.normal(0, s, 3)
        i["gyro_bias"] = np.zeros_like(i["gyro_bias"]); i["accel_bias"] = np.zeros_like(i["accel_bias"])
    for l in st["lidars"]:
        l["q"], l["t"], l["toff"] = _perturb_q(l["q"], rng, s), l["t"] + rng.normal(0, s, 3), 0.0
    return st


def zero_measurements(W):
    return dict(cams=[np.zeros((c["obs_first"][-1], 2)) for c in W["cams"]],
                accel=[np.zeros((i["t"].size, 3)) for i in W["imus"]], gyro=[np.zeros((i["t"].size, 3)) for i in W["imus"]],
                lidar=[np.zeros(l["t"].size) for l in W["lidars"]])


def build_spec(W, state, meas, unit_weights=False, with_mest=True, with_motion=True):
    """calibrator.buildProblem in batch form.  Handles recorded in W['handles'] for result extraction."""
    cfg = W["cfg"]
    sp = S.ProblemSpec()
    H = dict()
    H["pose"] = sp.add_spline(SPLINE_ORDER, W["pose_knots"], state["pose_coeffs"], True, S.HELPER_GROUP_ID)
    if W["imus"]:
        H["gravity"] = sp.add_dv(S.DV_EUCLIDEAN_DIRECTION, state["gravity"], True, 1.0, S.HELPER_GROUP_ID)
    H["boards"] = []
    for b, (q, t) in enumerate(state["boards"]):
        hq = sp.add_dv(S.DV_ROTATION_QUATERNION, q, b != 0, 1.0, S.HELPER_GROUP_ID)
        ht = sp.add_dv(S.DV_EUCLIDEAN_POINT, t, b != 0, 1.0, S.HELPER_GROUP_ID)
        H["boards"].append((hq, ht))
    H["imus"] = []
    for k, imu in enumerate(W["imus"]):
        si = state["imus"][k]
        hg = sp.add_spline(SPLINE_ORDER, imu["bias_knots"], si["gyro_bias"], True, S.HELPER_GROUP_ID)
        ha = sp.add_spline(SPLINE_ORDER, imu["bias_knots"], si["accel_bias"], True, S.HELPER_GROUP_ID)
        hq = sp.add_dv(S.DV_ROTATION_QUATERNION, si["q"], not imu["reference"], 1.0, S.HELPER_GROUP_ID)
        hr = sp.add_dv(S.DV_EUCLIDEAN_POINT, si["r"], not imu["reference"], 1.0, S.HELPER_GROUP_ID)
        H["imus"].append(dict(gyro_bias=hg, accel_bias=ha, q=hq, r=hr))
    H["cams"] = []
    for k, cam in enumerate(W["cams"]):
        sc = state["cams"][k]
        hq = sp.add_dv(S.DV_ROTATION_QUATERNION, sc["q"], True, 1.0, S.HELPER_GROUP_ID)
        ht = sp.add_dv(S.DV_EUCLIDEAN_POINT, sc["t"], True, 1.0, S.HELPER_GROUP_ID)
        hd = sp.add_dv(S.DV_SCALAR, [sc["toff"]], cfg.time_calibration, 1.0, S.CALIBRATION_GROUP_ID)
        hp = sp.add_dv(S.DV_PROJECTION, sc["proj"], False, 1.0, S.CALIBRATION_GROUP_ID)
        hk = sp.add_dv(S.DV_DISTORTION, sc["dk"], False, 1.0, S.CALIBRATION_GROUP_ID) if sc["dk"] is not None else -1
        hc = sp.add_camera(cam["model"], cam["dist"], hp, hk)
        H["cams"].append(dict(q=hq, t=ht, toff=hd, proj=hp, dist=hk, camera=hc))
    H["lidars"] = []
    for k, lid in enumerate(W["lidars"]):
        sl = state["lidars"][k]
        hd = sp.add_dv(S.DV_SCALAR, [sl["toff"]], cfg.time_calibration, 1.0, S.HELPER_GROUP_ID)
        hq = sp.add_dv(S.DV_ROTATION_QUATERNION, sl["q"], True, 1.0, S.CALIBRATION_GROUP_ID)
        ht = sp.add_dv(S.DV_EUCLIDEAN_POINT, sl["t"], True, 1.0, S.CALIBRATION_GROUP_ID)
        H["lidars"].append(dict(toff=hd, q=hq, t=ht))
    # ---- error terms: cameras, then per IMU accel / gyro / bias motion, then LiDAR
    pts = W["board_points"]
    for k, cam in enumerate(W["cams"]):
        h = H["cams"][k]
        n = int(cam["obs_first"][-1])
        p = np.concatenate([pts[idx] for idx in cam["corner_idx"]], 0) if n else np.zeros((0, 3))
        sinv = np.full(n, 1.0 if unit_weights else 1.0 / cfg.reproj_sigma)
        mest, mp = (S.MEST_NONE, 0.0)
        if with_mest and cfg.blake_zisserman_cam > 0:
            mest, mp = S.MEST_BLAKE_ZISSERMAN, S.blake_zisserman_epsilon(int(cfg.blake_zisserman_cam))
        sp.add_reprojection_batch(h["camera"], H["pose"], h["q"], h["t"], h["toff"], cam["obs_time"],
                                  [H["boards"][b][0] for b in cam["obs_board"]], [H["boards"][b][1] for b in cam["obs_board"]],
                                  cam["obs_first"], meas["cams"][k], p, sinv, cfg.time_padding, mest, mp)
    for k, imu in enumerate(W["imus"]):
        h = H["imus"][k]
        n = imu["t"].size
        ma = (S.MEST_HUBER, cfg.huber_accel) if with_mest and cfg.huber_accel > 0 else (S.MEST_NONE, 0.0)
        mg = (S.MEST_HUBER, cfg.huber_gyro) if with_mest and cfg.huber_gyro > 0 else (S.MEST_NONE, 0.0)
        sp.add_accelerometer_batch(H["pose"], h["accel_bias"], -1, H["gravity"], h["q"], h["r"], imu["t"], meas["accel"][k],
                                   np.full(n, 1.0 if unit_weights else 1.0 / cfg.accel_sigma), ma[0], ma[1])
        sp.add_gyroscope_batch(H["pose"], h["gyro_bias"], -1, h["q"], imu["t"], meas["gyro"][k],
                               np.full(n, 1.0 if unit_weights else 1.0 / cfg.gyro_sigma), mg[0], mg[1])
        if cfg.bias_motion_terms and with_motion:
            sp.add_motion_term(h["gyro_bias"], np.eye(3) / (1.9393e-5 ** 2), 1)
            sp.add_motion_term(h["accel_bias"], np.eye(3) / (3.0e-3 ** 2), 1)
    if cfg.pose_motion_terms and with_motion:
        sp.add_motion_term(H["pose"], np.diag([1e-6] * 3 + [1e-5] * 3), 2)
    # LiDAR terms are added plane by plane (addLiDARErrorTerms loops `for idx, obs in enumerate(self.targetObs)`,
    # sensors_and_targets.py:274-305), each plane's inliers in scan order
    for k, lid in enumerate(W["lidars"]):
        h = H["lidars"][k]
        lm = cfg.lidar_mest if (with_mest and cfg.lidar_mest) else (S.MEST_NONE, 0.0)
        for b in range(cfg.n_boards):
            idx = np.nonzero(lid["board"] == b)[0]
            if idx.size == 0:
                continue
            sp.add_lidar_batch(H["pose"], h["toff"], h["q"], h["t"], lid["t"][idx], lid["dir"][idx], meas["lidar"][k][idx],
                               np.full(idx.size, 1.0 if unit_weights else 1.0 / cfg.lidar_sigma),
                               [H["boards"][b][0]] * idx.size, [H["boards"][b][1]] * idx.size, cfg.time_padding, lm[0], lm[1])
    return sp, H


def synthesize(W, make_api, noise=True):
    """Measurements = forward model at the ground truth (+ noise).  `make_api()` returns a fresh backend object
    (HipApi on the GPU box, the oracle adapter in CPU tests)."""
    cfg = W["cfg"]
    spec, _ = build_spec(W, truth_state(W), zero_measurements(W), unit_weights=True, with_mest=False, with_motion=False)
    api = make_api()
    spec.emit(api)
    api.finalize()
    _, e = api.evaluate_error(False, True)
    rng = np.random.Generator(np.random.PCG64(cfg.seed + 1234))
    meas = zero_measurements(W)
    row = 0
    for k, cam in enumerate(W["cams"]):
        n = int(cam["obs_first"][-1])
        meas["cams"][k] = e[row:row + 2 * n].reshape(n, 2).copy()          # e_out = -(0 - yhat)
        row += 2 * n
    for k, imu in enumerate(W["imus"]):
        n = imu["t"].size
        meas["accel"][k] = -e[row:row + 3 * n].reshape(n, 3); row += 3 * n   # e_out = -(pred - 0)
        meas["gyro"][k] = -e[row:row + 3 * n].reshape(n, 3); row += 3 * n
    for k, lid in enumerate(W["lidars"]):
        for b in range(cfg.n_boards):
            idx = np.nonzero(lid["board"] == b)[0]
            meas["lidar"][k][idx] = -e[row:row + idx.size]
            row += idx.size
    assert row == e.size
    if noise:
        for k in range(len(W["cams"])):
            meas["cams"][k] += rng.normal(0, cfg.pixel_noise, meas["cams"][k].shape)
        for k in range(len(W["imus"])):
            meas["accel"][k] += rng.normal(0, cfg.accel_sigma, meas["accel"][k].shape)
            meas["gyro"][k] += rng.normal(0, cfg.gyro_sigma, meas["gyro"][k].shape)
        for k in range(len(W["lidars"])):
            meas["lidar"][k] += rng.normal(0, cfg.lidar_sigma, meas["lidar"][k].shape)
    return meas


def make_problem(cfg, make_api, noise=True):
    """World + measurements + the spec at the perturbed initial guess.  Returns (spec, handles, world, meas)."""
    W = make_world(cfg)
    meas = synthesize(W, make_api, noise)
    spec, H = build_spec(W, initial_state(W), meas)
    return spec, H, W, meas
