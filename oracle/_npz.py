"""Deterministic .npz writer for the golden fixtures (TEST INFRASTRUCTURE).

``numpy.savez_compressed`` stamps every zip member with the current time, so regenerating a fixture with identical
contents still changes the file's bytes (and its sha256 in tests/golden/meta.json). This writer fixes the member
timestamps and order: same arrays in -> same bytes out, and a fresh capture run is a no-op diff."""
import io
import zipfile

import numpy as np


def save(path, **arrays):
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED, compresslevel=6) as zf:
        for name in sorted(arrays):
            buf = io.BytesIO()
            np.lib.format.write_array(buf, np.asanyarray(arrays[name]), allow_pickle=False)
            info = zipfile.ZipInfo(name + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            info.external_attr = 0o644 << 16
            zf.writestr(info, buf.getvalue())
