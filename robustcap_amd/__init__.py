"""robustcap_amd -- MI355X-native (gfx950) implementation of RobustCap's sig_mp per-frame inference path.

Public surface (mirrors the reference for this path only):
    robustcap_amd.net.sig_mp.Net            <- net/sig_mp.py:Net (forward_online / reset_states / load_state_dict)
    robustcap_amd.body.ParametricModel      <- articulate/model.py:ParametricModel (FK / IK / landmark skinning)
    robustcap_amd.body.r6d_to_rotation_matrix <- articulate/math/angular.py
    robustcap_amd.smplify.smplify_runner    <- net/smplify/run.py (pre-check, L-BFGS optimiser, update mask)
    robustcap_amd.dist                      <- sequence sharding over GPUs + final gather (new)
    robustcap_amd.synth                     <- seeded synthetic weights / body / 60 fps inputs (new)
"""
__version__ = "0.1.0"
