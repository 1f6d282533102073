"""``Net`` -- the reference's sig_mp call surface on top of librobustcap_hip.so (MI355X / gfx950).

Drop-in for the inference half of ``net/sig_mp.py:23-274``:

    net = Net()                         # Net(body=..., batch=B) for the batched API
    net.load_state_dict(sd); net.eval()
    net.gravityc = Tcw[:3, :3] @ [0, -1, 0]           (evaluate.py:73)
    pose, tran = net.forward_online(j2dc[33,3], accc[6,3], oric[6,3,3], first_tran=None, first_frame=False)
    net.reset_states()

plus what the reference lacks: ``forward_batch`` (B bodies per call, outputs stay on the device) and
``forward_sequence`` ([B, T, ...] in one call, no host round trips). Row b of a batch equals the reference's
``forward_online`` run alone on sequence b.

Nothing is computed in Python or torch here: tensors are only allocated, made contiguous and handed to the C ABI
as raw device pointers on torch's current HIP stream. Without the HIP library this module fails at import of
``_lib.load()`` -- there is no CPU path.
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from .. import body as _body
from .. import config as cfg

_PARAM_ATTRS = {"use_flat_floor", "use_vision_updater", "use_imu_updater", "live", "conf_range", "tran_filter_num",
                "contact_threshold", "distrance_threshold", "height_threhold", "update_vision_freq", "use_reproj_opt",
                "smooth"}


class _ModuleView:
    """Read-only view of one node of the reference's module tree (net.rnn2, net.rnn2.rnn, net.rnn2.init_net[0] ...)."""

    def __init__(self, net, prefix):
        self.__dict__["_net"], self.__dict__["_prefix"] = net, prefix

    def _keys(self):
        return [k for k, _ in cfg.state_dict_spec() if k.startswith(self._prefix + ".")]

    def __getattr__(self, name):
        full = self._prefix + "." + name
        sd = self._net._sd_cpu
        if any(k == full for k, _ in cfg.state_dict_spec()):
            if full not in sd:
                raise AttributeError(f"{full}: not loaded yet")
            return torch.from_numpy(sd[full])
        if any(k.startswith(full + ".") for k, _ in cfg.state_dict_spec()):
            return _ModuleView(self._net, full)
        raise AttributeError(f"no parameter or sub-module {full!r}")

    def __getitem__(self, i):                       # nn.Sequential indexing: init_net[0], init_net[2], init_net[4]
        return self.__getattr__(str(int(i)))

    def __setattr__(self, name, value):
        raise AttributeError("read-only view: change weights through Net.load_state_dict")

    def state_dict(self):
        from collections import OrderedDict
        n = len(self._prefix) + 1
        return OrderedDict((k[n:], torch.from_numpy(self._net._sd_cpu[k])) for k in self._keys() if k in self._net._sd_cpu)

    def parameters(self):
        return iter(self.state_dict().values())

    def __repr__(self):
        return f"<view of {self._prefix}: {len(self._keys())} tensors>"


class Net:
    # class attributes callers read or poke on the reference (net/sig_mp.py:27-45)
    hidden_size = 512
    conf_range = (0.7, 0.8)
    contact_threshold = 0.7
    smooth = 1
    use_flat_floor = True
    use_reproj_opt = False
    use_vision_updater = True
    use_imu_updater = True
    name = "sig_mp"
    gravityc = torch.tensor([-0.0029, 0.9980, -0.0273])
    imu_num = 6
    height_threhold = 0.15
    distrance_threshold = 10
    tran_filter_num = 0.05
    live = False
    update_vision_freq = 30

    def __init__(self, body=None, batch=1, device="cuda"):
        """body: dict(J, v_template, weights, parent) (robustcap_amd.synth.make_body / body.load_smpl_pickle);
        None loads the reference's default ``models/SMPL_male.pkl`` (config.paths.smpl_file)."""
        self.__dict__["_ready"] = False
        self._lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RobustcapLibraryError("robustcap_amd.Net runs on a HIP device only (no CPU fallback)")
        self.batch = int(batch)
        live_ctor = bool(type(self).live)                                  # sig_mp.py:91-93
        self._ctx = C.c_void_p()
        _lib.check(None, self._lib.rc_create(self.batch, int(live_ctor), C.byref(self._ctx)), "rc_create")
        if body is None:
            path = os.path.join("models", "SMPL_male.pkl")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found; pass body=... (the SMPL pickle is an external download)")
            body = _body.load_smpl_pickle(path)
        _body.set_body(self._ctx, body)
        if live_ctor:
            self.__dict__["conf_range"] = (0.85, 0.9)
            self.__dict__["tran_filter_num"] = 0.01
        self._sd_cpu = {}
        self._loaded = False
        self._gravity_key = None
        self.__dict__["_ready"] = True
        self._push_params()

    # ------------------------------------------------------------------------------------------ attribute pokes
    def __setattr__(self, key, value):
        self.__dict__[key] = value
        if self.__dict__.get("_ready") and key in _PARAM_ATTRS:
            self._push_params()

    def _push_params(self):
        p = _lib.RcParams()
        _lib.check(self._ctx, self._lib.rc_get_params(self._ctx, C.byref(p)), "rc_get_params")
        p.conf_lo, p.conf_hi = float(self.conf_range[0]), float(self.conf_range[1])
        p.contact_threshold = float(self.contact_threshold)
        p.distance_threshold = float(self.distrance_threshold)
        p.height_threshold = float(self.height_threhold)
        p.tran_filter_num = float(self.tran_filter_num)
        p.use_flat_floor = int(bool(self.use_flat_floor))
        p.use_vision_updater = int(bool(self.use_vision_updater))
        p.use_imu_updater = int(bool(self.use_imu_updater))
        p.live = int(bool(self.live))
        p.update_vision_freq = int(self.update_vision_freq)
        p.use_reproj_opt = int(bool(self.use_reproj_opt))
        p.smooth = float(self.smooth)
        _lib.check(self._ctx, self._lib.rc_set_params(self._ctx, C.byref(p)), "rc_set_params")
        if self.__dict__.get("_live_on"):               # kernel arguments are baked into the captured frame
            _lib.check(self._ctx, self._lib.rc_live_end(self._ctx), "rc_live_end")
            self.__dict__["_live_on"] = False

    def _sync_gravity(self):
        g = self.gravityc                                                   # instance attr, else the class attr
        seen = self.__dict__.get("_gravity_seen")                           # same tensor object, not modified in place since
        if seen is not None and seen[0] is g and isinstance(g, torch.Tensor) and seen[1] == g._version:
            return
        mark = (g, g._version) if isinstance(g, torch.Tensor) else None
        g = torch.as_tensor(g, dtype=torch.float32).detach().cpu().reshape(-1, 3)
        key = g.numpy().tobytes()
        if key != self._gravity_key:
            g = g.expand(self.batch, 3).contiguous() if g.shape[0] == 1 else g.contiguous()
            if g.shape[0] != self.batch:
                raise ValueError(f"gravityc must be [3] or [{self.batch}, 3]")     # nothing recorded: the next call raises again
            _lib.check(self._ctx, self._lib.rc_set_gravity(self._ctx, _lib.ptr(g)), "rc_set_gravity")
            self.__dict__["_gravity_key"] = key
        self.__dict__["_gravity_seen"] = mark                               # only a value that reached the device is remembered

    # ------------------------------------------------------------------------------------- torch.nn.Module-like
    def to(self, device=None, *a, **k):
        if device is not None and torch.device(device).type != "cuda":
            raise _lib.RobustcapLibraryError("robustcap_amd.Net runs on a HIP device only (no CPU fallback)")
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict, strict=True):
        """Same key names / shapes as the reference ``Net.state_dict()``. Values: torch tensors or
        numpy arrays. Repacks to the kernel layout and uploads.

        Aliasing contract (unlike torch, which copies into its parameters): float32 CPU tensors / contiguous arrays are kept BY
        REFERENCE until the next load, so that a later partial (``strict=False``) load can re-send the rest without this object
        holding a second 254 MB copy. Editing such a tensor in place after loading it therefore shows up in the NEXT load that
        re-sends it -- pass a copy if the caller's tensor is going to change. (Other dtypes / devices are converted, and the
        converted copy is what is kept.)"""
        want = dict(cfg.state_dict_spec())
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Net: missing {missing[:4]} unexpected {unexpected[:4]}")
        new = {}
        for k, shape in want.items():
            if k not in state_dict:
                continue
            v = state_dict[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(v.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
            new[k] = np.ascontiguousarray(v, dtype=np.float32)      # float32 CPU tensors / arrays: a view of the caller's memory
        # The library packs from the tensors passed since its last finalize and keeps no host copy: a partial (non-strict)
        # load goes on top of the tensors of the previous loads, which are kept here BY REFERENCE (no second copy).
        self._sd_cpu.update(new)
        for k, v in self._sd_cpu.items():
            _lib.check(self._ctx, self._lib.rc_load_weight(self._ctx, k.encode(), v.ctypes.data_as(C.c_void_p), v.size), "rc_load_weight")
        _lib.check(self._ctx, self._lib.rc_finalize_weights(self._ctx), "rc_finalize_weights")
        self.__dict__["_live_on"] = False          # the library dropped its captured frame (it held the old weight pointers)
        self.__dict__["_loaded"] = True
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ------------------------------------------------------------------------------------------ module introspection
    # The reference's Net is a torch.nn.Module (net/sig_mp.py:23); its callers on the path only load weights and call it (evaluate.py:56-93,
    # live_server.py:62-67). What introspects a module finds the same names here: state_dict() / named_parameters() / parameters() in the
    # reference's key order and the attribute tree net.rnn2.linear1.weight, net.rnn4.rnn.weight_hh_l1, net.rnn2.init_net[4].bias ... as
    # read-only views of the loaded tensors (the weights live packed in device memory; an edit goes through load_state_dict).
    def state_dict(self):
        from collections import OrderedDict
        return OrderedDict((k, torch.from_numpy(self._sd_cpu[k])) for k, _ in cfg.state_dict_spec() if k in self._sd_cpu)

    def named_parameters(self, prefix="", recurse=True):
        for k, v in self.state_dict().items():
            yield (prefix + ("." if prefix else "") + k, v)

    def parameters(self, recurse=True):
        for _, v in self.named_parameters():
            yield v

    def train(self, mode=True):
        if mode:
            raise _lib.RobustcapLibraryError("robustcap_amd.Net is the inference path only (net/sig_mp.py:301-560, the training half, is out of scope)")
        return self

    def __getattr__(self, name):
        # (only reached when normal lookup fails) sub-module views: rnn2 ... rnn8
        sd = self.__dict__.get("_sd_cpu")
        if sd is not None and name.startswith("rnn") and any(k.startswith(name + ".") for k, _ in cfg.state_dict_spec()):
            return _ModuleView(self, name)
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    def reset_states(self, rows=None):
        """net/sig_mp.py:95-104. rows: optional bool/uint8 mask [batch] (batched API)."""
        m = None
        if rows is not None:
            m = torch.as_tensor(rows).to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self._ctx, self._lib.rc_reset(self._ctx, _lib.ptr(m), _lib.stream_ptr()), "rc_reset")
        if m is not None:
            torch.cuda.current_stream().synchronize()                       # keep `m` alive until the kernel ran

    # ---------------------------------------------------------------------------------------------- hot path
    def _prep(self, t, shape):
        return t.to(device=self.device, dtype=torch.float32).reshape(shape).contiguous()

    def _prep_rows(self, t, B, T, width):
        """[B, T, ...] input of a sequence call as (tensor, row stride in floats) without a copy when it already is float32
        on the device with contiguous frames -- e.g. a slice ``x[:, a:b]`` of a longer contiguous sequence tensor (the C ABI
        takes the row stride; only a frame's ``width`` floats have to be contiguous and frames ``width`` apart)."""
        if t.is_cuda and t.dtype == torch.float32 and t.dim() >= 3 and t.shape[0] == B and t.shape[1] == T:
            inner = tuple(t.shape[2:])
            st = t.stride()
            expect, ok = 1, True
            for d in range(t.dim() - 1, 1, -1):                       # the frame itself contiguous
                ok = ok and (st[d] == expect or t.shape[d] == 1)
                expect *= t.shape[d]
            if ok and expect == width and (st[1] == width or T == 1) and (st[0] >= T * width or B == 1) and inner:
                return t, int(st[0]) if B > 1 else T * width
        t = self._prep(t, (B, T, width))
        return t, T * width

    @torch.no_grad()
    def forward_batch(self, j2dc, accc, oric, first_tran=None, first_frame=False):
        """B bodies, one frame. Inputs [B,33,3], [B,6,3], [B,6,3,3]; returns device tensors pose [B,24,3,3], tran [B,3]."""
        B = self.batch
        self._sync_gravity()
        j2dc, accc, oric = self._prep(j2dc, (B, 33, 3)), self._prep(accc, (B, 6, 3)), self._prep(oric, (B, 6, 3, 3))
        ft = None if first_tran is None else self._prep(first_tran, (B, 3))
        pose = torch.empty(B, 24, 3, 3, device=self.device)
        tran = torch.empty(B, 3, device=self.device)
        rc = self._lib.rc_step(self._ctx, _lib.ptr(j2dc), _lib.ptr(accc), _lib.ptr(oric), _lib.ptr(ft),
                               _lib.RC_FLAG_FIRST_FRAME if first_frame else 0, _lib.ptr(pose), _lib.ptr(tran), _lib.stream_ptr())
        _lib.check(self._ctx, rc, "rc_step")
        self.__dict__["_keep"] = (j2dc, accc, oric, ft)                    # inputs must outlive the async launches
        return pose, tran

    @torch.no_grad()
    def forward_online(self, j2dc, accc, oric, first_tran=None, first_frame=False):
        """net/sig_mp.py:113-274: one body, one frame; returns CPU tensors like the reference (L274).
        With ``net.use_graph = True`` (live loops) the frame is one hipGraph replay incl. the host copies."""
        if self.batch != 1:
            raise ValueError("forward_online is the batch-1 call; use forward_batch")
        d = self.__dict__
        if d.get("use_graph"):
            # the live loop's steady state (live_server.py:40-48 hands over CPU float32 tensors every 16 ms): straight to rc_live_step --
            # no no_grad scope, no batch views, outputs allocated in their final shape (the Python side of a frame: ~10 -> ~8 us, of which 3 are the two
            # torch.empty calls the reference's return-fresh-tensors contract needs)
            if (d.get("_live_on") and first_tran is None and type(j2dc) is torch.Tensor and type(accc) is torch.Tensor and type(oric) is torch.Tensor
                    and j2dc.dtype is torch.float32 and accc.dtype is torch.float32 and oric.dtype is torch.float32
                    and j2dc.is_cpu and accc.is_cpu and oric.is_cpu
                    and j2dc.is_contiguous() and accc.is_contiguous() and oric.is_contiguous()
                    and j2dc.numel() == 99 and accc.numel() == 18 and oric.numel() == 54):
                self._sync_gravity()
                pose, tran = torch.empty(24, 3, 3), torch.empty(3)
                rc = self._lib.rc_live_step(self._ctx, j2dc.data_ptr(), accc.data_ptr(), oric.data_ptr(), None,
                                            _lib.RC_FLAG_FIRST_FRAME if first_frame else 0, pose.data_ptr(), tran.data_ptr())
                if rc:
                    _lib.check(self._ctx, rc, "rc_live_step")
                return pose, tran
            p, t = self.forward_live(j2dc, accc, oric, first_tran, first_frame)
            return p[0], t[0]
        pose, tran = self.forward_batch(j2dc, accc, oric, first_tran, first_frame)
        return pose[0].cpu(), tran[0].cpu()

    @torch.no_grad()
    def forward_live(self, j2dc, accc, oric, first_tran=None, first_frame=False):
        """Streaming step for all rows with HOST tensors in and out (live_server.py:40-48): one captured hipGraph
        per frame (14 kernels, or 11 when no row can need a transition step; inputs and outputs in pinned host memory). Returns CPU tensors pose [B,24,3,3], tran [B,3]."""
        B = self.batch
        self._sync_gravity()
        if not self.__dict__.get("_live_on"):
            torch.cuda.synchronize()
            _lib.check(self._ctx, self._lib.rc_live_begin(self._ctx), "rc_live_begin")
            self.__dict__["_live_on"] = True
        def host(x, n):       # CPU float32 contiguous tensors (what live callers hand over) pass through untouched
            if isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.device.type == "cpu" and x.is_contiguous() and x.numel() == n:
                return x
            return torch.as_tensor(x, dtype=torch.float32).cpu().reshape(n).contiguous()
        j2dc, accc, oric = host(j2dc, B * 99), host(accc, B * 18), host(oric, B * 54)
        ft = None if first_tran is None else host(first_tran, B * 3)
        pose, tran = torch.empty(B, 24, 3, 3), torch.empty(B, 3)
        rc = self._lib.rc_live_step(self._ctx, j2dc.data_ptr(), accc.data_ptr(), oric.data_ptr(), None if ft is None else ft.data_ptr(),
                                    _lib.RC_FLAG_FIRST_FRAME if first_frame else 0, pose.data_ptr(), tran.data_ptr())
        _lib.check(self._ctx, rc, "rc_live_step")
        return pose, tran

    def fusion_state(self):
        """int32 [batch, 5]: last_tran set, len(floor_y), first_reach, update_vision_count, deferred updater step pending."""
        t = torch.empty(self.batch, 5, dtype=torch.int32)
        _lib.check(self._ctx, self._lib.rc_get_fusion_state(self._ctx, _lib.ptr(t), _lib.stream_ptr()), "rc_get_fusion_state")
        return t

    def live_stats(self):
        """(frames replayed from the lean seven-launch capture, frames on the full captures) of forward_live so far."""
        a, b = C.c_int64(0), C.c_int64(0)
        _lib.check(self._ctx, self._lib.rc_get_live_stats(self._ctx, C.byref(a), C.byref(b)), "rc_get_live_stats")
        return a.value, b.value

    def live_prestep_stats(self):
        """(pre-steps enqueued so far, whether the live session carries the pre-step programs): rc_get_live_prestep -- the recurrent
        halves of the next frame's layer steps, computed while a paced (60 fps) caller leaves the device idle between two frames."""
        a, b = C.c_int64(0), C.c_int32(0)
        _lib.check(self._ctx, self._lib.rc_get_live_prestep(self._ctx, C.byref(a), C.byref(b)), "rc_get_live_prestep")
        return a.value, bool(b.value)

    def live_spin_stats(self):
        """(frames that started from a first kernel launched ahead of them, such kernels sent away or timed out): rc_get_live_spin (RC_LIVE_SPIN=1)."""
        a, b = C.c_int64(0), C.c_int64(0)
        _lib.check(self._ctx, self._lib.rc_get_live_spin(self._ctx, C.byref(a), C.byref(b)), "rc_get_live_spin")
        return a.value, b.value

    def live_replayed(self):
        """Lean live frames whose own device-side check found them off the lean plan (a transition step or an init_net trigger the host-side
        mirror in rc_live_step did not foresee): they change nothing and are replayed on the full capture. Expected: 0."""
        a = C.c_int64(0)
        _lib.check(self._ctx, self._lib.rc_get_live_replayed(self._ctx, C.byref(a)), "rc_get_live_replayed")
        return a.value

    @torch.no_grad()
    def forward_sequence(self, j2dc, accc, oric, first_tran=None, first_frame=False):
        """The evaluate.py frame loop (evaluate.py:75-83) for B sequences of T frames in one call.
        Inputs [B,T,33,3], [B,T,6,3], [B,T,6,3,3]; returns device tensors pose [B,T,24,3,3], tran [B,T,3]."""
        B = self.batch
        T = j2dc.shape[1]
        if T == 0:                                                          # (the reference's loop over no frames: nothing happens)
            return torch.empty(B, 0, 24, 3, 3, device=self.device), torch.empty(B, 0, 3, device=self.device)
        self._sync_gravity()
        (j2dc, rs_j), (accc, rs_a), (oric, rs_o) = self._prep_rows(j2dc, B, T, 99), self._prep_rows(accc, B, T, 18), self._prep_rows(oric, B, T, 54)
        ft = None if first_tran is None else self._prep(first_tran, (B, 3))
        pose = torch.empty(B, T, 24, 3, 3, device=self.device)
        tran = torch.empty(B, T, 3, device=self.device)
        rc = self._lib.rc_sequence(self._ctx, T, _lib.ptr(j2dc), rs_j, _lib.ptr(accc), rs_a, _lib.ptr(oric), rs_o, _lib.ptr(ft),
                                   _lib.RC_FLAG_FIRST_FRAME if first_frame else 0, _lib.ptr(pose), T * 216, _lib.ptr(tran), T * 3,
                                   _lib.stream_ptr())
        _lib.check(self._ctx, rc, "rc_sequence")
        self.__dict__["_keep"] = (j2dc, accc, oric, ft)
        return pose, tran

    @staticmethod
    def default_gemm_mode(total_rows):
        """The library's default product arithmetic for a workload of ``total_rows`` bodies (rc_default_gemm_mode: split-bf16
        products from 48 rows). Sharded runs pass their TOTAL row count here and pin every shard's context with
        ``set_gemm_mode``, so that a row's bits do not depend on the number of ranks it was split over."""
        return bool(_lib.load().rc_default_gemm_mode(int(total_rows)))

    def set_gemm_mode(self, split):
        """Product arithmetic of every GEMM of this context (rc_set_gemm_mode): False = fp32 MFMA (fma chains), True =
        split-bf16 partial products with fp32 accumulation (default for batch >= 48). Results are bitwise reproducible
        across batch sizes and shards WITHIN one mode."""
        _lib.check(self._ctx, self._lib.rc_set_gemm_mode(self._ctx, int(bool(split))), "rc_set_gemm_mode")
        self.__dict__["_live_on"] = False

    @property
    def gemm_mode(self):
        return int(self._lib.rc_get_gemm_mode(self._ctx))

    def set_sequence_mode(self, enabled=True, min_frames=8, force=False):
        """Scheduling of ``forward_sequence`` (rc_set_sequence_mode). ``enabled`` (the default): calls of at least
        ``min_frames`` frames run on the per-row-cursor wavefront engine when the launch plan's cost estimate beats the
        frame-stepped launches (``force``: whenever they are long enough). Outputs and states are bitwise the same either way."""
        mode = 0 if not enabled else (2 if force else 1)
        _lib.check(self._ctx, self._lib.rc_set_sequence_mode(self._ctx, mode, int(min_frames)), "rc_set_sequence_mode")

    def sequence_stats(self):
        """(frames run by the wavefront engine, frames run frame-stepped, ticks launched) since construction."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self._ctx, self._lib.rc_get_sequence_stats(self._ctx, C.byref(a), C.byref(b), C.byref(c)), "rc_get_sequence_stats")
        return a.value, b.value, c.value

    # ------------------------------------------------------------------------------------------- introspection
    def lstm_step(self, net, x, rows=None):
        """One step f(i, x) of a sub-net on the context's state (net/sig_mp.py:126-129); for component tests."""
        spec = {n: (i, h, o) for n, i, h, o in cfg.NETS}[net]
        x = self._prep(x, (self.batch, spec[0]))
        m = None if rows is None else torch.as_tensor(rows).to(device=self.device, dtype=torch.uint8).contiguous()
        y = torch.zeros(self.batch, spec[2], device=self.device)
        _lib.check(self._ctx, self._lib.rc_lstm_step(self._ctx, net.encode(), _lib.ptr(x), _lib.ptr(m), _lib.ptr(y), _lib.stream_ptr()), "rc_lstm_step")
        torch.cuda.current_stream().synchronize()
        return y

    def get_state(self, net):
        """(h, c) of a sub-net as CPU tensors [2, batch, H]."""
        H = {n: h for n, _, h, _ in cfg.NETS}[net]
        h = torch.empty(2, self.batch, H)
        c = torch.empty(2, self.batch, H)
        _lib.check(self._ctx, self._lib.rc_get_state(self._ctx, net.encode(), _lib.ptr(h), _lib.ptr(c), _lib.stream_ptr()), "rc_get_state")
        return h, c

    def get_trace(self):
        """int32 [batch, 8] branch trace of the last frame (see rc_get_trace in the header)."""
        t = torch.empty(self.batch, 8, dtype=torch.int32)
        _lib.check(self._ctx, self._lib.rc_get_trace(self._ctx, _lib.ptr(t), _lib.stream_ptr()), "rc_get_trace")
        return t

    def gemm_kernel_name(self):
        """Name of the kernel that carries the path's FLOPs in this context (what a rocprofv3 kernel summary lists): the
        shared-weight kernel of the LSTM layer steps where it has run, else the wide-tile kernel of the context's arithmetic."""
        try:
            lds, other = self.launch_stats()
        except AttributeError:                                              # (an older build of the library under RC_LIB_PATH: A/B runs)
            lds, other = 0, 0
        if lds > 0:
            return "rc_gemm_lds_kernel"
        try:
            w = C.c_int64()
            _lib.check(self._ctx, self._lib.rc_get_launch_stats_w32(self._ctx, C.byref(w)), "rc_get_launch_stats_w32")
            if w.value > 0:                                                 # contexts of 33-64 rows: the layer steps run as one-reader launches
                return "rc_gemm_split48_w32_kernel"
        except AttributeError:
            pass
        return "rc_gemm_split_kernel" if self.gemm_mode else "rc_gemm_kernel"

    def launch_stats(self):
        """(launches of the shared-weight kernel rc_gemm_lds_kernel, launches of the other wide-tile kernels) since construction."""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(self._ctx, self._lib.rc_get_launch_stats(self._ctx, C.byref(a), C.byref(b)), "rc_get_launch_stats")
        return a.value, b.value

    def set_resident(self, enable=True, workgroups=0):
        """Resident layer-step kernel of the wavefront engine (include/robustcap_hip.h: rc_set_resident): one launch carries the LSTM layer
        steps and linear1 layers of every tick of a planned forward_sequence call. Bitwise the stream engine's results; off by default."""
        _lib.check(self._ctx, self._lib.rc_set_resident(self._ctx, 1 if enable else 0, int(workgroups)), "rc_set_resident")

    def resident_stats(self):
        """(segments run on the resident kernel, aborted ones) since construction."""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(self._ctx, self._lib.rc_get_resident_stats(self._ctx, C.byref(a), C.byref(b)), "rc_get_resident_stats")
        return a.value, b.value

    def gemm_timing(self, enable):
        """0 off, 1 (True) every gate-GEMM launch, 2 only the launches of the wide-tile kernels, 3 only those of rc_gemm_lds_kernel."""
        _lib.check(self._ctx, self._lib.rc_gemm_timing(self._ctx, int(enable)), "rc_gemm_timing")

    def gemm_timing_read(self):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self._ctx, self._lib.rc_gemm_timing_read(self._ctx, C.byref(ms), C.byref(n)), "rc_gemm_timing_read")
        return ms.value, n.value

    def gemm_timing_busy(self):
        """ms with at least one of the timed launches running (launches on two streams overlap); after gemm_timing_read()."""
        ms = C.c_double()
        _lib.check(self._ctx, self._lib.rc_gemm_timing_busy(self._ctx, C.byref(ms)), "rc_gemm_timing_busy")
        return ms.value

    def __del__(self):
        ctx = self.__dict__.get("_ctx")
        if ctx:
            self._lib.rc_destroy(ctx)
            self.__dict__["_ctx"] = None

