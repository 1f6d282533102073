"""Live streaming front-end: the wire formats and the per-packet loop of the reference's ``live_server.py`` so that
its own detector / IMU processes (live_detector.py, live_demo_sync.py) can drive the MI355X path unchanged.

Wire formats (ASCII text, except the IMU bridge's binary datagram):
  IMU bridge -> sync process, UDP (live_demo_sync.py:262-268): 8 N float32 values [t x N | q x 4N | a x 3N] (parse_imu_packet).
  detector -> server, UDP 127.0.0.1:9999 (live_detector.py:57-61):  "uv#ori#acc#RCM" where each field is a
      comma-separated list of floats: uv 33x3 (x/z, y/z, visibility), ori 6x3x3, acc 6x3, RCM 3x3.
  server -> Unity, TCP 127.0.0.1:8888 (live_server.py:57-59):  "%g,"-joined 72 axis-angle values '#' 3 translation
      values, terminated by '$'.
Loop (live_server.py:30-60): the first packet only sets gravity = RCM [0,-1,0]; every later packet is one
``forward_online`` (``first_frame=True`` on the first of them), the root rotation and translation are rotated back by
RCM^T, the translation is made relative to its first value, rotations go out as axis-angle.

Host-side plumbing only (sockets, text); the frame itself is ``Net.forward_live`` = one hipGraph replay.
"""
import socket

import numpy as np
import torch

from . import body as _body


def parse_floats(text):
    """live_server.py:17-22 convert_from_str."""
    return np.asarray([float(v) for v in text.split(",")])


def parse_detector_packet(data):
    """bytes -> (uv [33,3], ori [6,3,3], acc [6,3], RCM [3,3]) float32 tensors (live_server.py:33-43)."""
    uv, ori, acc, rcm = data.decode().split("#")
    f = lambda s, shape: torch.from_numpy(parse_floats(s)).reshape(shape).float()
    return f(uv, (33, 3)), f(ori, (6, 3, 3)), f(acc, (6, 3)), f(rcm, (3, 3))


def format_detector_packet(uv, ori, acc, rcm):
    """(live_detector.py:57-60) what the detector process sends; numpy float32 str() per element."""
    j = lambda a: ",".join(str(i) for i in np.asarray(a, np.float32).reshape(-1))
    return (j(uv) + "#" + j(ori) + "#" + j(acc) + "#" + j(rcm)).encode()


def parse_imu_packet(data, n_imus):
    """One UDP datagram of the IMU bridge (live_demo_sync.py:262-268 ``get_from_udp``): 8 N little-endian float32 values
    ``[t x N | q x 4N | a x 3N]`` = per-sensor time stamps, orientation quaternions (N x 4, sensor order, w first as the Xsens Dot
    SDK sends them) and accelerations (N x 3). Returns (t list[float], q [N, 4], a [N, 3]); raises ValueError on a datagram of
    another length (the reference would fail in ``reshape``)."""
    n = int(n_imus)
    buf = np.frombuffer(data, np.float32)
    if buf.size != 8 * n:
        raise ValueError(f"IMU packet: {buf.size} float32 values, expected {8 * n} for {n} sensors")
    buf = buf.copy()
    return buf[:n].tolist(), torch.from_numpy(buf[n:5 * n].reshape(n, 4)), torch.from_numpy(buf[5 * n:].reshape(n, 3))


def format_imu_packet(t, q, a):
    """What the bridge sends (the inverse of parse_imu_packet): bytes of the float32 values [t | q | a]."""
    return np.concatenate([np.asarray(t, np.float32).reshape(-1), np.asarray(q, np.float32).reshape(-1), np.asarray(a, np.float32).reshape(-1)]).tobytes()


def format_unity_packet(pose_axis_angle, tran):
    """(live_server.py:57-58) -> bytes."""
    return (",".join("%g" % v for v in pose_axis_angle) + "#" + ",".join("%g" % v for v in tran) + "$").encode("utf8")


class LiveSession:
    """State of one live run: gravity from the first packet, first_frame on the first pose packet, start-relative
    translation. ``net`` needs forward_online(uv, acc, ori, first_frame=...) and a settable ``gravityc`` (a
    robustcap_amd Net with ``live = True`` / ``use_graph = True``, or any stand-in in tests)."""

    def __init__(self, net, device="cuda"):
        self.net, self.device = net, device
        self.rcm = None
        self.start_tran = None

    def handle(self, data):
        """One UDP packet in -> the Unity packet out (None for the very first, gravity-only packet)."""
        uv, ori, acc, rcm = parse_detector_packet(data)
        if self.rcm is None:                                            # live_server.py:32-35
            self.rcm = rcm
            self.net.gravityc = torch.matmul(rcm, torch.tensor([0.0, -1.0, 0.0]).unsqueeze(-1)).squeeze(-1)
            return None
        pose, tran = self.net.forward_online(uv, acc, ori, first_frame=self.start_tran is None)
        pose, tran = pose.clone(), tran.clone()
        pose[0] = self.rcm.T.matmul(pose[0])                            # live_server.py:49-51
        tran = self.rcm.T.matmul(tran.unsqueeze(-1)).squeeze(-1)
        if self.start_tran is None:
            self.start_tran = tran.clone()
        tran = tran - self.start_tran
        aa = _body.rotation_matrix_to_axis_angle(pose, self.device).cpu().view(-1)   # live_server.py:55
        return format_unity_packet(aa.tolist(), tran.tolist())


def run_live_server(net, server_ip="127.0.0.1", unity_ip="127.0.0.1", udp_port=9999, tcp_port=8888, max_packets=None, device="cuda"):
    """live_server.py:24-60 with the same ports and blocking behaviour. ``max_packets`` bounds the loop (tests)."""
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind((unity_ip, tcp_port))
    srv.listen(1)
    conn, _ = srv.accept()
    s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    s.bind((server_ip, udp_port))
    sess = LiveSession(net, device)
    n = 0
    try:
        while max_packets is None or n < max_packets:
            data, _ = s.recvfrom(4000000)
            out = sess.handle(data)
            if out is not None:
                conn.send(out)
            n += 1
    finally:
        conn.close(), srv.close(), s.close()
    return n
