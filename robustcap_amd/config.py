"""Constants of the sig_mp hot path.

Values are the data constants of the reference's ``config.py`` (L97-101) and the class attributes of
``net/sig_mp.py:Net`` (L27-45). Only what the per-frame path reads is kept; dataset paths, live-camera
settings and split lists are out of scope.
"""

# config.py:97 -- velocity scale of rnn3's output (m/s * 3 / 60 fps)
vel_scale = 3
# config.py:98
tran_offset = (0.0, 0.25, 5.0)
# config.py:99 -- SMPL vertex ids standing in for the 33 MediaPipe landmarks
mp_mask = (332, 2809, 2800, 455, 6260, 3634, 3621, 583, 4071, 45, 3557, 1873, 4123, 1652, 5177, 2235, 5670,
           2673, 6133, 2319, 5782, 2746, 6191, 3138, 6528, 1176, 4662, 3381, 6727, 3387, 6787, 3226, 6624)
# config.py:100 -- SMPL vertex ids the 6 IMUs sit on
vi_mask = (1961, 5424, 1176, 4662, 411, 3021)
# config.py:101 -- SMPL joint ids of the 6 IMUs (L-elbow, R-elbow, L-knee, R-knee, head, pelvis)
ji_mask = (18, 19, 4, 5, 15, 0)

# standard SMPL kinematic tree (kintree_table[0] of the official pickle, articulate/model.py:38-39)
smpl_parent = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)

# sync_mp3d (net/sig_mp.py:287-299): landmark row -> SMPL joint id that overwrites it
mp_joint_override = {11: 16, 12: 17, 13: 18, 14: 19, 15: 20, 16: 21, 23: 1, 24: 2, 25: 4, 26: 5, 27: 7, 28: 8}

# MediaPipe landmark ids whose confidence the smplify residual zeroes (temporal_smplify.py:92)
smplify_ignored_landmarks = (1, 2, 3, 4, 5, 6, 7, 8, 9, 31, 32)

NUM_IMU = 6
NUM_KP = 33
NUM_JOINT = 24

# (name, input, hidden, output) of the six sub-nets, net/sig_mp.py:52-81
NETS = (
    ("rnn2", 72, 512, 69),
    ("rnn3", 141, 512, 3),
    ("rnn4", 171, 1280, 69),
    ("rnn6", 240, 1024, 3),
    ("rnn7", 141, 512, 144),
    ("rnn8", 141, 512, 2),
)
NET_INDEX = {n[0]: i for i, n in enumerate(NETS)}
# rnn2.init_net: Linear(69,512) ReLU Linear(512,1024) ReLU Linear(1024,2048)  (rnn.py:195-201)
INIT_NET = ((69, 512), (512, 1024), (1024, 2048))

# 2 * MACs of one pass of all six nets (linear1 + 2 LSTM layers + linear2), SURVEY.md section 8(d)
FLOPS_PER_BODY_FRAME = 121_379_840
# of which the six linear2 layers (2 * H * out each); at batch > 16 they run on rc_gemm_small_kernel, everything else
# (linear1 + both LSTM layers of all six sub-nets) on rc_gemm_kernel
FLOPS_LINEAR2_PER_BODY_FRAME = 2 * (512 * 69 + 512 * 3 + 1280 * 69 + 1024 * 3 + 512 * 144 + 512 * 2)


def state_dict_spec():
    """[(key, shape)] of every tensor in the reference ``Net.state_dict()`` in torch's order."""
    spec = []
    for name, nin, h, nout in NETS:
        for l in (0, 1):
            spec += [(f"{name}.rnn.weight_ih_l{l}", (4 * h, h)), (f"{name}.rnn.weight_hh_l{l}", (4 * h, h)),
                     (f"{name}.rnn.bias_ih_l{l}", (4 * h,)), (f"{name}.rnn.bias_hh_l{l}", (4 * h,))]
        spec += [(f"{name}.linear1.weight", (h, nin)), (f"{name}.linear1.bias", (h,)),
                 (f"{name}.linear2.weight", (nout, h)), (f"{name}.linear2.bias", (nout,))]
        if name == "rnn2":
            for i, (a, b) in zip((0, 2, 4), INIT_NET):
                spec += [(f"{name}.init_net.{i}.weight", (b, a)), (f"{name}.init_net.{i}.bias", (b,))]
    return spec

# the twelve LSTM layer steps alone (2 layers x [B, 2H] x [2H, 4H] per sub-net): what rc_gemm_lds_kernel computes
FLOPS_LSTM_PER_BODY_FRAME = 2 * sum(2 * (2 * H) * (4 * H) for H in (512, 512, 1280, 1024, 512, 512))
