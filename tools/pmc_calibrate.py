#!/usr/bin/env python3
"""Run the three calibration streams (1, 4, 16 bytes per lane) so that rocprofv3 --pmc FETCH_SIZE can be compared
with a known byte count:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_calibrate.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslam3_cpp_fork_amd import _lib  # noqa: E402
from orbslam3_cpp_fork_amd.extractor import ORBextractor  # noqa: E402

ex = ORBextractor(1000)
NB = 1 << 30  # 1 GiB: beyond the 256 MiB Infinity Cache
for w in (16, 4, 1):
    rc = _lib.lib().orbx_debug_calib_read(ex._h, NB, w)
    print("width", w, "bytes", NB, "rc", rc)
