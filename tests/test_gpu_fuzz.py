"""A bounded slice of the randomised parity run (tests/fuzz_common.py) inside the driver's -m gpu run: ~45 s of
frames with the default settings, then ~25 s in which every ray pool of the wavefront integrator is traced as PACKETS
however small or spread out it is (the settings a real frame only reaches at full size), with several launch
groups per frame.  The long runs (tens of thousands of frames) are tools/fuzz_parity.py's."""
import os

import pytest

import fuzz_common

pytestmark = pytest.mark.gpu


def test_fuzz_default_settings(rwr, orc, ref_loader):
    with rwr.Context(0) as ctx:
        n, n_path, n_dormant, worst = fuzz_common.run(rwr, orc, ref_loader, ctx, seed=20260, seconds=45.0)
    assert n >= 30 and n_path >= 5 and worst <= 1e-4, (n, n_path, worst)


def test_fuzz_forced_packets_and_small_groups(rwr, orc, ref_loader):
    saved = {k: os.environ.get(k) for k in ("RWR_WF_GROUP", "RWR_WF_PACKET_FILL", "RWR_WF_PACKET_EXTENT", "RWR_WF_MIN_PACKET_POOLS")}
    os.environ.update({"RWR_WF_GROUP": "5", "RWR_WF_PACKET_FILL": "0", "RWR_WF_PACKET_EXTENT": "1e30", "RWR_WF_MIN_PACKET_POOLS": "0"})
    try:
        with rwr.Context(0) as ctx:        # the tunables are read when the context is created
            n, n_path, _, worst = fuzz_common.run(rwr, orc, ref_loader, ctx, seed=777, seconds=25.0, path_fraction=0.9, many_samples=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert n >= 10 and n_path >= 8 and worst <= 1e-4, (n, n_path, worst)
