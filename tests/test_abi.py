"""The C-ABI library loads and exports every symbol include/robustcap_hip.h declares (CPU: no compute calls)."""
import ctypes as C
import os
import re

import pytest

from robustcap_amd import _lib

HEADER = open(_lib.HEADER_PATH).read()
DECLARED = sorted(set(re.findall(r"^(?:int|const char\*)\s+(rc_\w+)\s*\(", HEADER, flags=re.M)))


def test_header_declares_the_documented_surface():
    for name in ("rc_create", "rc_destroy", "rc_last_error", "rc_load_weight", "rc_finalize_weights", "rc_set_body",
                 "rc_set_gravity", "rc_reset", "rc_step", "rc_sequence", "rc_r6d_to_rotmat", "rc_ik_r", "rc_fk_bone",
                 "rc_body_fk", "rc_lstm_step", "rc_reproj_residual", "rc_get_state", "rc_get_trace"):
        assert name in DECLARED
    # every entry point cites the reference interface it replaces
    assert HEADER.count("net/sig_mp.py") >= 8 and "articulate/model.py" in HEADER and "net/smplify" in HEADER


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "HIP library not built: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in DECLARED if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == DECLARED, "ctypes prototypes out of sync with the header"


def test_no_torch_types_in_the_abi():
    code = re.sub(r"/\*.*?\*/", "", HEADER[HEADER.index('extern "C"'):], flags=re.S)      # declarations without comments
    assert "at::" not in code and "torch" not in code.lower() and "std::" not in code and "Tensor" not in code
    assert "void* stream" in code and "const float*" in code                                  # plain pointers + stream


def test_default_params_match_reference_class_attributes():
    lib = _lib.load()
    p = _lib.RcParams()
    assert lib.rc_default_params(0, C.byref(p)) == 0
    assert (p.conf_lo, p.conf_hi, p.tran_filter_num) == (0.7, 0.8, 0.05)          # net/sig_mp.py:28,40
    assert abs(p.contact_threshold - 0.7) < 1e-7 and p.distance_threshold == 10 and abs(p.height_threshold - 0.15) < 1e-7
    assert (p.use_flat_floor, p.use_vision_updater, p.use_imu_updater, p.live, p.update_vision_freq) == (1, 1, 1, 0, 30)
    assert lib.rc_default_params(1, C.byref(p)) == 0
    assert (p.conf_lo, p.conf_hi, p.tran_filter_num, p.live) == (0.85, 0.9, 0.01, 1)  # net/sig_mp.py:91-93


def test_errors_are_codes_not_exceptions():
    import torch
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.rc_create(0, 0, C.byref(ctx)) == -1                                 # bad batch
    assert b"batch" in lib.rc_last_error(None)
    if not torch.cuda.is_available():                                              # no device: loud failure, no fallback
        assert lib.rc_create(1, 0, C.byref(ctx)) == -2
        assert lib.rc_last_error(None)
        from robustcap_amd.net.sig_mp import Net
        with pytest.raises(_lib.RobustcapLibraryError):
            Net(body={"J": None}, batch=1)
    assert lib.rc_step(None, None, None, None, None, 0, None, None, None) == -1
    assert lib.rc_destroy(None) == 0


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is usable from C: the header compiles as strict C99 and a C program links against the library and
    calls host-only entry points (no GPU needed: parameter defaults, error strings, the float64 L-BFGS)."""
    import shutil
    import subprocess
    from robustcap_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include "robustcap_hip.h"
static double quad(void* user, const double* x, double* g, int64_t n) {
    double f = 0.0; (void)user;
    for (int64_t i = 0; i < n; ++i) { const double d = x[i] - (double)(i + 1); f += d * d; g[i] = 2.0 * d; }
    return f;
}
int main(void) {
    rc_params p;
    if (rc_default_params(1, &p) != 0 || p.conf_lo != 0.85 || p.live != 1) return 2;
    if (rc_destroy(NULL) != 0) return 3;
    double x[4] = {0, 0, 0, 0}, losses[32];
    int32_t it = 0, ev = 0;
    if (rc_lbfgs_minimize(quad, NULL, 4, x, 1.0, 20, 25, 100, 1e-7, 1e-9, &it, &ev, losses, 32) != 0) return 4;
    for (int i = 0; i < 4; ++i) if (x[i] < i + 1 - 1e-6 || x[i] > i + 1 + 1e-6) return 5;
    printf("ok %d %d\n", it, ev);
    return 0;
}
''')
    inc = os.path.dirname(_lib.HEADER_PATH)
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(exe),
                    _lib.LIB_PATH, "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok ")
