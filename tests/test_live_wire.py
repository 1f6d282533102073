"""Wire formats of the live front-end (CPU): packet parse/format round trip and the Unity string of live_server.py."""
import numpy as np
import pytest
import torch

from robustcap_amd import live


def test_detector_packet_round_trip():
    rng = np.random.default_rng(0)
    uv, ori, acc, rcm = (rng.standard_normal(s).astype(np.float32) for s in ((33, 3), (6, 3, 3), (6, 3), (3, 3)))
    pkt = live.format_detector_packet(uv, ori, acc, rcm)
    assert pkt.count(b"#") == 3 and len(pkt.split(b"#")[0].split(b",")) == 99
    u2, o2, a2, r2 = live.parse_detector_packet(pkt)
    assert np.array_equal(u2.numpy(), uv) and np.array_equal(o2.numpy(), ori)         # shortest float32 repr is exact
    assert np.array_equal(a2.numpy(), acc) and np.array_equal(r2.numpy(), rcm)


def test_unity_packet_is_the_reference_expression():
    pose = [0.1234567, -1e-5, 3.0] * 24
    tran = [0.5, -0.25, 1e-7]
    ref = ",".join(["%g" % v for v in pose]) + "#" + ",".join(["%g" % v for v in tran]) + "$"   # live_server.py:57-58
    assert live.format_unity_packet(pose, tran) == ref.encode("utf8")


def test_session_logic_with_a_stand_in_net(monkeypatch):
    class FakeNet:
        gravityc = None
        calls = []

        def forward_online(self, uv, acc, ori, first_frame=False):
            self.calls.append(first_frame)
            return torch.eye(3).repeat(24, 1, 1), torch.tensor([1.0, 2.0, 3.0]) * len(self.calls)

    monkeypatch.setattr(live._body, "rotation_matrix_to_axis_angle", lambda r, device="cpu": torch.zeros(24, 3))
    net = FakeNet()
    sess = live.LiveSession(net, device="cpu")
    rcm = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    pkt = live.format_detector_packet(np.zeros((33, 3)), np.tile(np.eye(3), (6, 1, 1)), np.zeros((6, 3)), rcm)
    assert sess.handle(pkt) is None                                          # first packet: gravity only
    assert torch.allclose(net.gravityc, torch.tensor([1.0, 0.0, 0.0]))       # RCM [0,-1,0]
    out1, out2 = sess.handle(pkt), sess.handle(pkt)
    assert net.calls == [True, False]                                        # first_frame only on the first pose packet
    assert out1.endswith(b"#0,0,0$")                                         # translation relative to its first value
    t2 = [float(v) for v in out2[:-1].split(b"#")[1].split(b",")]
    assert np.allclose(t2, rcm.T @ np.array([1.0, 2.0, 3.0]))                # RCM^T (tran_2 - tran_1)


def test_imu_udp_packet_round_trip_and_layout():
    """live_demo_sync.py:262-268: [t x N | q x 4N | a x 3N] float32, 32 N bytes per datagram."""
    from robustcap_amd import live
    rng = np.random.default_rng(3)
    for n in (1, 6):
        t, q, a = rng.random(n).astype(np.float32), rng.standard_normal((n, 4)).astype(np.float32), rng.standard_normal((n, 3)).astype(np.float32)
        data = live.format_imu_packet(t, q, a)
        assert len(data) == 32 * n
        raw = np.frombuffer(data, np.float32)                               # the reference's own slicing
        assert np.array_equal(raw[:n], t) and np.array_equal(raw[n:5 * n].reshape(n, 4), q) and np.array_equal(raw[5 * n:].reshape(n, 3), a)
        t2, q2, a2 = live.parse_imu_packet(data, n)
        assert t2 == t.tolist() and torch.equal(q2, torch.from_numpy(q)) and torch.equal(a2, torch.from_numpy(a))
        assert q2.dtype == torch.float32 and q2.shape == (n, 4) and a2.shape == (n, 3)
    with pytest.raises(ValueError):
        live.parse_imu_packet(b"\0" * 28, 1)
