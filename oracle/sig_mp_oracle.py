"""CPU oracle for the sig_mp per-frame path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A from-scratch PyTorch-CPU restatement of what the reference computes per frame, written batched (B bodies
advance together, per-row masks replace the reference's ``.item()`` branches) so that row b of a batch equals
the reference's ``forward_online`` run alone on sequence b. Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg may import this module; the product (``robustcap_amd``) never does.

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against vectors captured from the
reference itself (``oracle/capture_reference.py`` -> ``tests/golden``), including eight scripted
multi-regime sequences with a branch trace. Unpinned: ``rotation_matrix_to_axis_angle`` (the reference calls
OpenCV 4.2 ``cv2.Rodrigues``, absent here; restated from the Rodrigues formula, validated by round trip).

Reference lines restated (all under /root/reference):
  net/sig_mp.py:113-274  Net.forward_online        -> OracleNet.forward_batch / forward_online
  net/sig_mp.py:85-104   state + reset_states      -> OracleNet.reset_states
  net/sig_mp.py:126-129  f(i, x) single LSTM step  -> OracleRNN.step
  net/sig_mp.py:277-299  get_bbox_scale, sync_mp3d -> normalize_keypoints, OracleBody.landmarks
  articulate/utils/torch/rnn.py:92-133,174-219     -> OracleRNN (same state_dict key names)
  articulate/math/angular.py:249-264, 221-233      -> r6d_to_rotation_matrix, axis_angle_to_rotation_matrix
  articulate/math/spatial.py:104-221               -> OracleBody.inverse_kinematics_R / bone_fk
  articulate/model.py:78-93, 209-241               -> OracleBody.__init__ / forward_kinematics
  net/smplify/losses.py:6-12, 36-37, 43-46 + temporal_smplify.py:198-220 -> reprojection_residual
"""
import math

import numpy as np
import torch

from robustcap_amd import config as C

F32 = torch.float32


# ------------------------------------------------------------------------------------------------ math (L0)
def normalize(x):
    return x / x.norm(dim=-1, keepdim=True)


def r6d_to_rotation_matrix(r6d):
    """[..., 6] -> [N, 3, 3]; columns are (c0, c1, c0 x c1); NaN -> 0 (angular.py:249-264)."""
    r6d = r6d.reshape(-1, 6)
    a, b = r6d[:, 0:3], r6d[:, 3:6]
    c0 = normalize(a)
    c1 = normalize(b - (c0 * b).sum(dim=1, keepdim=True) * c0)
    c2 = torch.linalg.cross(c0, c1, dim=1)
    r = torch.stack((c0, c1, c2), dim=-1)
    return torch.where(torch.isnan(r), torch.zeros_like(r), r)


def axis_angle_to_rotation_matrix(a):
    """[N,3] -> [N,3,3] by Rodrigues' formula; zero vector -> identity (angular.py:221-233)."""
    a = a.reshape(-1, 3)
    angle = a.norm(dim=1, keepdim=True)
    axis = a / angle
    axis = torch.where(torch.isfinite(axis), axis, torch.zeros_like(axis))
    x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
    zero = torch.zeros_like(x)
    K = torch.stack((zero, -z, y, z, zero, -x, -y, x, zero), dim=1).view(-1, 3, 3)
    c, s = angle.cos().view(-1, 1, 1), angle.sin().view(-1, 1, 1)
    eye = torch.eye(3, dtype=a.dtype).expand(a.shape[0], 3, 3)
    return c * eye + (1 - c) * (axis.view(-1, 3, 1) @ axis.view(-1, 1, 3)) + s * K


def rotation_matrix_to_axis_angle(R):
    """[N,3,3] -> [N,3]. The reference loops ``cv2.Rodrigues`` (angular.py:236-246; opencv-python-headless 4.2.0.34,
    requirements.txt:7). OpenCV is absent here, so this restates the matrix -> vector branch of its ``cvRodrigues2``
    (modules/calib3d/src/calibration.cpp in 4.2) step by step, in float64 like OpenCV's own arithmetic:

      1. entries outside [-100, 100] or NaN              -> zero vector
      2. R <- U V^T of the SVD of R                      (nearest orthonormal matrix)
      3. r = (R21 - R12, R02 - R20, R10 - R01);  s = sqrt(|r|^2 / 4);  c = clamp((tr R - 1) / 2, -1, 1);  theta = acos(c)
      4. s < 1e-5:  c > 0 -> 0;  else r = sqrt(max((R_ii + 1) / 2, 0)) with the signs of R01, R02 (and the R12 fix-up
         when r_x is the smallest component), scaled to length theta
      5. otherwise  r * theta / (2 s)
      6. result rounded to the input depth (float32).

    PARITY UNPINNED: no OpenCV build is available to run against; the restatement is cross-checked against scipy's
    independent log map (tests) and by round trip with the pinned axis_angle_to_rotation_matrix."""
    Rn = R.reshape(-1, 3, 3).double().numpy()
    N = Rn.shape[0]
    out = np.zeros((N, 3))
    bad = ~np.all(np.isfinite(Rn) & (np.abs(Rn) < 100.0), axis=(1, 2))          # cv::checkRange(-100, 100), NaN fails
    ok = np.where(~bad)[0]
    if ok.size:
        U, _, Vt = np.linalg.svd(Rn[ok])
        Q = U @ Vt
        r = np.stack((Q[:, 2, 1] - Q[:, 1, 2], Q[:, 0, 2] - Q[:, 2, 0], Q[:, 1, 0] - Q[:, 0, 1]), axis=1)
        s = np.sqrt((r * r).sum(1) * 0.25)
        c = np.clip((Q[:, 0, 0] + Q[:, 1, 1] + Q[:, 2, 2] - 1.0) * 0.5, -1.0, 1.0)
        theta = np.arccos(c)
        res = np.zeros_like(r)
        big = s >= 1e-5
        res[big] = r[big] * (theta[big] / (2.0 * s[big]))[:, None]
        flip = np.where(~big & ~(c > 0))[0]
        for k in flip:                                                           # theta ~ pi (rare): scalar code like OpenCV's
            q = Q[k]
            rx = math.sqrt(max((q[0, 0] + 1) * 0.5, 0.0))
            ry = math.sqrt(max((q[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if q[0, 1] < 0 else 1.0)
            rz = math.sqrt(max((q[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if q[0, 2] < 0 else 1.0)
            if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((q[1, 2] > 0) != (ry * rz > 0)):
                rz = -rz
            n = math.sqrt(rx * rx + ry * ry + rz * rz)
            res[k] = np.array([rx, ry, rz]) * (theta[k] / n) if n > 0 else 0.0
        out[ok] = res
    return torch.from_numpy(out.astype(np.float32))


def rotation_angle_deg(Ra, Rb):
    """geodesic angle between rotation matrices, float64 atan2 form."""
    D = Ra.reshape(-1, 3, 3).double().transpose(1, 2) @ Rb.reshape(-1, 3, 3).double()
    v = torch.stack((D[:, 2, 1] - D[:, 1, 2], D[:, 0, 2] - D[:, 2, 0], D[:, 1, 0] - D[:, 0, 1]), dim=1) * 0.5
    c = (D[:, 0, 0] + D[:, 1, 1] + D[:, 2, 2] - 1) * 0.5
    return torch.rad2deg(torch.atan2(v.norm(dim=1), c))


def lerp_rows(a, b, k64):
    """a*(1-k)+b*k with k a float64 per-row weight: (1-k) is formed in double, then each factor is rounded to
    float32 for the multiply -- what ``tensor * python_float`` does (general.py:24, sig_mp.py:163-164)."""
    w1 = (1.0 - k64).to(F32).unsqueeze(1)
    w2 = k64.to(F32).unsqueeze(1)
    return a * w1 + b * w2


def bbox_scale(uv):
    """max(width, height) of the keypoint bounding box, [B,33,>=2] -> [B] (sig_mp.py:277-284)."""
    u, v = uv[..., 0], uv[..., 1]
    return torch.maximum(u.max(dim=-1).values - u.min(dim=-1).values, v.max(dim=-1).values - v.min(dim=-1).values)


def normalize_keypoints(kp):
    """scale xy by the bbox, then make every row except 23 relative to row 23 (sig_mp.py:150-152)."""
    kp = kp.clone()
    kp[..., :2] = kp[..., :2] / bbox_scale(kp).view(-1, 1, 1)
    hip = kp[:, 23:24, :2].clone()
    kp[:, 24:, :2] = kp[:, 24:, :2] - hip
    kp[:, :23, :2] = kp[:, :23, :2] - hip
    return kp


def gmof(x, sigma):
    x2 = x * x
    s2 = sigma * sigma
    return (s2 * x2) / (s2 + x2)


# ------------------------------------------------------------------------------------------------ body (L1)
class OracleBody:
    """SMPL-format body restricted to what the path consumes: 24 joints + the 33 landmark vertices."""

    def __init__(self, body, vertex_ids=C.mp_mask):
        self.parent = [int(p) for p in body["parent"]]
        J = torch.as_tensor(body["J"], dtype=F32)
        vt = torch.as_tensor(body["v_template"], dtype=F32)
        self.j_rest = J - J[:1]                                               # model.py:87
        self.v_rest = (vt - J[:1])[list(vertex_ids)]
        self.w = torch.as_tensor(body["weights"], dtype=F32)[list(vertex_ids)]   # [V,24]
        par = torch.tensor([0] + self.parent[1:])
        self.par = par
        self.bone = self.j_rest - self.j_rest[par]                            # spatial.py:148-167
        self.bone[0] = self.j_rest[0]
        self.override = sorted(C.mp_joint_override.items())

    def inverse_kinematics_R(self, Rg):
        Rg = Rg.reshape(-1, 24, 3, 3)
        loc = Rg[:, self.par[1:]].transpose(-1, -2) @ Rg[:, 1:]
        return torch.cat((Rg[:, :1], loc), dim=1)

    def forward_kinematics_R(self, Rl):
        Rl = Rl.reshape(-1, 24, 3, 3)
        G = [Rl[:, 0]]
        for i in range(1, 24):
            G.append(G[self.parent[i]] @ Rl[:, i])
        return torch.stack(G, dim=1)

    def bone_to_joint(self, pb):
        pb = pb.reshape(pb.shape[0], -1, 3)
        P = [pb[:, 0]]
        for i in range(1, 24):
            P.append(P[self.parent[i]] + pb[:, i])
        return torch.stack(P, dim=1)

    def bone_fk(self, Rg):
        """joint positions from GLOBAL rotations and rest bone vectors, root at 0 (sig_mp.py:131-135)."""
        Rg = Rg.reshape(-1, 24, 3, 3)
        pb = (Rg[:, self.par[1:]] @ self.bone[1:].view(1, 23, 3, 1)).squeeze(-1)
        pb = torch.cat((torch.zeros(Rg.shape[0], 1, 3), pb), dim=1)
        return self.bone_to_joint(pb)

    def forward_kinematics(self, pose, tran):
        """local rotations [B,24,3,3] + root position -> (global rot, joints [B,24,3], vertices [B,V,3]);
        blend the 24 joint transforms per vertex, then apply (model.py:229-241)."""
        pose = pose.reshape(-1, 24, 3, 3)
        G, P = [pose[:, 0]], [torch.zeros(pose.shape[0], 3)]
        for i in range(1, 24):
            p = self.parent[i]
            G.append(G[p] @ pose[:, i])
            P.append((G[p] @ self.bone[i].view(1, 3, 1)).squeeze(-1) + P[p])
        G, P = torch.stack(G, dim=1), torch.stack(P, dim=1)
        t = P - (G @ self.j_rest.view(1, 24, 3, 1)).squeeze(-1)                 # model.py:235
        A = torch.cat((G, t.unsqueeze(-1)), dim=-1)                             # [B,24,3,4]
        Av = torch.einsum("vj,bjrc->bvrc", self.w, A)                           # model.py:236
        v = (Av[..., :3] @ self.v_rest.view(1, -1, 3, 1)).squeeze(-1) + Av[..., 3]
        tr = tran.view(-1, 1, 3)
        return G, P + tr, v + tr

    def landmarks(self, vert, joint):
        """sync_mp3d: 33 landmark vertices with 12 rows replaced by SMPL joints (sig_mp.py:287-299)."""
        j = vert.clone()
        for row, jid in self.override:
            j[:, row] = joint[:, jid]
        return j


def reprojection_residual(body, pose, tran, kp, K, sigma=100.0, ignored=C.smplify_ignored_landmarks):
    """smplify forward residual [T,33]: conf^2 * sum_xy gmof(K * (j33 / z) - kp) with the ignored landmarks'
    confidence zeroed (temporal_smplify.py:198-220, losses.py:36-37,43-46)."""
    _, joint, vert = body.forward_kinematics(pose, tran)
    j33 = body.landmarks(vert, joint)
    proj = (K @ (j33 / j33[..., 2:]).unsqueeze(-1)).squeeze(-1)[..., :2]
    conf = kp[..., 2].clone()
    conf[:, list(ignored)] = 0.0
    return conf * conf * gmof(proj - kp[..., :2], sigma).sum(dim=-1)


# ------------------------------------------------------------------------------------------------- nets (L2)
class OracleRNN(torch.nn.Module):
    """Linear -> ReLU -> 2-layer LSTM -> Linear; parameter names equal the reference's (rnn.py:92-133)."""

    def __init__(self, n_in, n_hidden, n_out, with_init=False):
        super().__init__()
        self.rnn = torch.nn.LSTM(n_hidden, n_hidden, 2)
        self.linear1 = torch.nn.Linear(n_in, n_hidden)
        self.linear2 = torch.nn.Linear(n_hidden, n_out)
        if with_init:                                                        # rnn.py:195-201
            self.init_net = torch.nn.Sequential(
                torch.nn.Linear(n_out, n_hidden), torch.nn.ReLU(),
                torch.nn.Linear(n_hidden, 2 * n_hidden), torch.nn.ReLU(),
                torch.nn.Linear(2 * n_hidden, 4 * n_hidden))
        self.n_hidden = n_hidden

    def step(self, x, h, c, rows=None):
        """advance rows ``rows`` (index tensor or None = all) of state (h, c) [2,B,H] in place; return y."""
        x1 = torch.relu(self.linear1(x)).unsqueeze(0)
        if rows is None:
            y, (hn, cn) = self.rnn(x1, (h, c))
            h.copy_(hn), c.copy_(cn)
        else:
            y, (hn, cn) = self.rnn(x1, (h[:, rows].contiguous(), c[:, rows].contiguous()))
            h[:, rows] = hn
            c[:, rows] = cn
        return self.linear2(y.squeeze(0))


class OracleNet(torch.nn.Module):
    """Batched restatement of ``Net`` (inference half). ``batch`` bodies share weights, each with its own state."""

    def __init__(self, body, batch=1, live=False):
        super().__init__()
        for name, nin, nh, nout in C.NETS:
            setattr(self, name, OracleRNN(nin, nh, nout, with_init=(name == "rnn2")))
        self.body = body if isinstance(body, OracleBody) else OracleBody(body)
        self.B = batch
        self.live = live
        self.conf_range = (0.85, 0.9) if live else (0.7, 0.8)               # sig_mp.py:28,91-93
        self.tran_filter_num = 0.01 if live else 0.05
        self.contact_threshold = 0.7
        self.distance_threshold = 10.0
        self.height_threshold = 0.15
        self.use_flat_floor = True
        self.use_vision_updater = True
        self.use_imu_updater = True
        self.use_reproj_opt = False
        self.smooth = 1
        self.update_vision_freq = 30
        self.gravityc = torch.tensor([-0.0029, 0.9980, -0.0273]).repeat(batch, 1)   # sig_mp.py:36
        self.update_vision_count = torch.zeros(batch, dtype=torch.long)      # class attrs: survive reset_states
        self.j_temp = torch.zeros(batch, 33, 3)
        self.trace = None
        self.reset_states()
        self.eval()

    @torch.no_grad()
    def reset_states(self, rows=None):
        B = self.B
        if rows is None:
            self.h = {n: torch.zeros(2, B, nh) for n, _, nh, _ in C.NETS}
            self.c = {n: torch.zeros(2, B, nh) for n, _, nh, _ in C.NETS}
            self.last_pfoot = torch.zeros(B, 2, 3)
            self.last_tran = torch.zeros(B, 3)
            self.has_last = torch.zeros(B, dtype=torch.bool)
            self.floor = torch.zeros(B, 11, 3)
            self.n_floor = torch.zeros(B, dtype=torch.long)
            self.first_reach = torch.ones(B, dtype=torch.bool)
        else:
            for n, *_ in C.NETS:
                self.h[n][:, rows] = 0
                self.c[n][:, rows] = 0
            self.has_last[rows] = False
            self.n_floor[rows] = 0
            self.first_reach[rows] = True

    def load_numpy_state_dict(self, sd):
        self.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}, strict=True)

    def _step(self, name, x, rows=None):
        if rows is not None and rows.numel() == 0:
            return torch.zeros(0, getattr(self, name).linear2.out_features)
        return getattr(self, name).step(x if rows is None else x[rows], self.h[name], self.c[name], rows)

    @torch.no_grad()
    def forward_online(self, j2dc, accc, oric, first_tran=None, first_frame=False):
        assert self.B == 1
        ft = None if first_tran is None else first_tran.view(1, 3)
        p, t = self.forward_batch(j2dc.view(1, 33, 3), accc.view(1, 6, 3), oric.view(1, 6, 3, 3), ft, first_frame)
        return p[0], t[0]

    @torch.no_grad()
    def forward_batch(self, j2dc, accc, oric, first_tran=None, first_frame=False):
        B, body = self.B, self.body
        lo, hi = self.conf_range
        j2dc, accc, oric = j2dc.reshape(B, 33, 3).float(), accc.reshape(B, 6, 3).float(), oric.reshape(B, 6, 3, 3).float()
        flat = lambda *xs: torch.cat([x.reshape(B, -1) for x in xs], dim=1)

        c64 = j2dc[:, :, 2].mean(dim=1).double()                              # L138 (python double compares)
        is_hi, is_mid = c64 >= hi, (c64 > lo) & (c64 < hi)
        vis = (c64 > lo) | first_frame                                        # L149
        Rcr = oric[:, 5]                                                      # L139
        accr = accc @ Rcr                                                     # L142
        orir = Rcr.transpose(1, 2).unsqueeze(1) @ oric                        # L143
        imu_r = flat(accr, orir)
        imu_c = flat(accc, oric)

        j3dr_i = self._step("rnn2", imu_r)                                    # L144
        vr = self._step("rnn3", flat(imu_r, j3dr_i))                          # L145

        n4 = torch.zeros(B, dtype=torch.long)
        n6 = torch.zeros(B, dtype=torch.long)
        j3dc = torch.zeros(B, 69)
        pc = torch.zeros(B, 3)
        rows_vis = vis.nonzero().flatten()
        if rows_vis.numel():
            j3dc[rows_vis] = self._step("rnn4", flat(imu_c, normalize_keypoints(j2dc)), rows_vis)   # L150-153
            n4[rows_vis] += 1
        x6 = flat(imu_c, j2dc, j3dc)
        if first_frame:                                                       # L155-156
            pc = self._step("rnn6", x6)
            n6 += 1
        rows_pc = (c64 > lo).nonzero().flatten()                              # L161 / L165
        if rows_pc.numel():
            pc[rows_pc] = self._step("rnn6", x6, rows_pc)
            n6[rows_pc] += 1
        j3dr_v = (j3dc.view(B, 23, 3) @ Rcr).reshape(B, 69)                   # L154
        k64 = (c64 - lo) / (hi - lo)                                          # L163
        j3dr = torch.where(is_hi.unsqueeze(1), j3dr_v,
                           torch.where(is_mid.unsqueeze(1), lerp_rows(j3dr_i, j3dr_v, k64), j3dr_i))

        x78 = flat(imu_r, j3dr)
        poseg6d = self._step("rnn7", x78)                                     # L169
        contact = torch.sigmoid(self._step("rnn8", x78))                      # L170

        poseg = r6d_to_rotation_matrix(poseg6d).view(B, 24, 3, 3)             # L173
        pose = body.inverse_kinematics_R(poseg)                               # L174
        pose[:, 0] = Rcr                                                      # L175

        reach = is_hi & self.first_reach & self.use_imu_updater               # L178-183
        rows_r = reach.nonzero().flatten()
        if rows_r.numel():
            hc = self.rnn2.init_net(j3dr[rows_r]).view(-1, 2, 2, 512).permute(1, 2, 0, 3)
            self.h["rnn2"][:, rows_r] = hc[0]
            self.c["rnn2"][:, rows_r] = hc[1]
            self.first_reach[rows_r] = False

        pfoot = body.bone_fk(poseg)[:, 10:12] @ Rcr.transpose(1, 2)           # L186
        cmax = contact.max(dim=1).values
        use_vel = (cmax < self.contact_threshold) | ~self.has_last            # L187
        v_vel = (Rcr @ vr.view(B, 3, 1)).view(B, 3) * C.vel_scale / 60        # L188
        foot = contact.argmax(dim=1)
        v_foot = (self.last_pfoot - pfoot)[torch.arange(B), foot]             # L190
        v = torch.where(use_vel.unsqueeze(1), v_vel, v_foot)
        tran = torch.where(self.has_last.unsqueeze(1), self.last_tran + v, v)  # L191-194

        kf = torch.clamp(k64, max=1.0)                                        # L197-199
        far = ((pc - tran).norm(dim=1) > self.distance_threshold) | (self.tran_filter_num > 1)
        fused = torch.where(far.unsqueeze(1), pc, lerp_rows(tran, pc, self.tran_filter_num * kf))
        tran = torch.where(is_hi.unsqueeze(1), fused, tran)                   # L196-203

        g = self.gravityc
        ft_given = first_tran is not None
        on_ground = cmax > self.contact_threshold

        def ground(i):
            return ((pfoot[:, i] + tran) * g).sum(dim=1, keepdim=True) * g

        sample = (self.n_floor < 11) & on_ground & is_hi                      # L208-214
        n_add = torch.zeros(B, dtype=torch.long)
        if self.use_flat_floor and not first_frame and not ft_given and sample.any():
            p0, p1 = ground(0), ground(1)
            pick = torch.where((p0.norm(dim=1) < p1.norm(dim=1)).unsqueeze(1), p1, p0)
            rows = sample.nonzero().flatten()
            self.floor[rows, self.n_floor[rows]] = pick[rows]
            self.n_floor[rows] += 1
            n_add[rows] = 1
        apply = (self.n_floor > 10) & on_ground                               # L215-221
        if self.use_flat_floor and apply.any():
            p0, p1 = ground(0), ground(1)
            m = self.floor[:, 5:11]
            mean = (((((m[:, 0] + m[:, 1]) + m[:, 2]) + m[:, 3]) + m[:, 4]) + m[:, 5]) / 6
            use1 = (p0.norm(dim=1) < p1.norm(dim=1)) & ((mean - p1).norm(dim=1) < self.height_threshold)
            use0 = ~use1 & ((mean - p0).norm(dim=1) < self.height_threshold)
            d = torch.where(use1.unsqueeze(1), mean - p1, torch.where(use0.unsqueeze(1), mean - p0, torch.zeros(B, 3)))
            tran = torch.where(apply.unsqueeze(1), tran + d, tran)
        if ft_given:                                                          # L222-225
            tran = first_tran.reshape(B, 3).float().clone()
        elif first_frame:
            tran = pc.clone()
        self.last_pfoot = pfoot                                               # L227
        self.has_last = torch.ones(B, dtype=torch.bool)

        if self.live:                                                         # L228-242
            refresh = self.update_vision_count == 0
        else:
            refresh = torch.ones(B, dtype=torch.bool)
        _, joint, vert = body.forward_kinematics(pose, tran)
        j_new = body.landmarks(vert, joint)
        j33 = torch.where(refresh.view(B, 1, 1), j_new, self.j_temp)
        if self.live and (self.use_reproj_opt or self.use_vision_updater):    # L228: the counter only moves inside this block
            self.j_temp = j33.clone()
            self.update_vision_count = torch.where(refresh, torch.full_like(self.update_vision_count, self.update_vision_freq),
                                                   self.update_vision_count - 1)

        if self.use_reproj_opt:                                               # L245-261: closed-form tran refinement
            m = (c64 > lo).unsqueeze(1)
            p, u2, v2 = j2dc[:, :, 2], j2dc[:, :, 0], j2dc[:, :, 1]
            jx, jy, jz = j33[..., 0], j33[..., 1], j33[..., 2]
            ax = (p / jz.pow(2)).sum(dim=1) + self.smooth
            bx = (p * (-jx / jz.pow(2) + u2 / jz)).sum(dim=1)
            by = (p * (-jy / jz.pow(2) + v2 / jz)).sum(dim=1)
            d1 = torch.stack((bx / ax, by / ax, torch.zeros(B)), dim=1)
            j1 = j33 + d1.unsqueeze(1)
            jx, jy, jz = j1[..., 0], j1[..., 1], j1[..., 2]
            az = (p * (jx.pow(2) + jy.pow(2)) / jz.pow(4)).sum(dim=1) + self.smooth
            bz = (p * ((jx / jz - u2) * jx / jz.pow(2) + (jy / jz - v2) * jy / jz.pow(2))).sum(dim=1)
            d2 = torch.stack((torch.zeros(B), torch.zeros(B), bz / az), dim=1)
            tran = torch.where(m, (tran + d1) + d2, tran)
            j33 = torch.where(m.unsqueeze(2), j1 + d2.unsqueeze(1), j33)

        upd = (c64 <= lo) & refresh & self.use_vision_updater                 # L264-271
        rows_u = upd.nonzero().flatten()
        if rows_u.numel():
            kp = j33 / j33[:, :, 2:]
            j3 = (joint[:, 1:] - joint[:, :1]).reshape(B, 69)
            self._step("rnn6", flat(imu_c, kp, j3), rows_u)
            self._step("rnn4", flat(imu_c, normalize_keypoints(kp)), rows_u)
            n6[rows_u] += 1
            n4[rows_u] += 1

        self.last_tran = tran.clone()                                         # L273
        self.trace = dict(c=c64, n4=n4, n6=n6, n_floor_add=n_add, n_floor=self.n_floor.clone(), reach=reach,
                          use_vel=use_vel, foot=foot, far=far & is_hi, contact=contact, pfoot=pfoot, j33=j33,
                          joint=joint, j3dr_i=j3dr_i, vr=vr, j3dc=j3dc, pc=pc, poseg6d=poseg6d,
                          count=self.update_vision_count.clone())
        return pose, tran


def build_oracle(weight_seed=0, body_seed=1, batch=1, live=False, state_dict=None, body=None):
    """OracleNet with the seeded synthetic assets (robustcap_amd.synth)."""
    from robustcap_amd import synth
    body = body if body is not None else synth.make_body(body_seed)
    net = OracleNet(body, batch=batch, live=live)
    net.load_numpy_state_dict(state_dict if state_dict is not None else synth.make_state_dict(weight_seed))
    return net
