"""bench.py's derived fields, without a GPU: the physical roofline fraction must be a fraction (<= 1) that follows from the
committed rocprofv3 PMC summaries by the formulas of DESIGN.md §7, as SCALAR fields (the driver's parser keeps scalars only),
and `parity_rows` must tell a bit-identical frame from one that differs in one float."""
import json
import os

import numpy as np
import pytest
from conftest import REPO

import bench


@pytest.mark.parametrize("scene,lds,kernel_ms,launches", [("cbox", True, 2.68, 1), ("scene1", True, 0.32, 1), ("bunny", False, 4.83, 1),
                                                          ("buddha_standin", False, 84.4, 1), ("dragon_standin", False, 611.0, 4)])
def test_physical_roofline_is_a_fraction_made_of_scalars(scene, lds, kernel_ms, launches):
    r = bench.physical_roofline(scene, "exact", lds, kernel_ms, launches)
    assert all(isinstance(v, (int, float, str)) for v in r.values()), r
    assert r["pmc_source"].startswith("profiles/r03_") and 0.0 < r["physical_frac"] <= 1.0
    prof = json.load(open(os.path.join(REPO, r["pmc_source"])))
    c, dv = prof["counters_mean_per_launch"], prof["derived"]
    t = kernel_ms * 1e-3
    if lds:
        assert r["physical_bound"] == "valu_lane_throughput"
        want = c["SQ_INSTS_VALU"] * launches / t / (1024 * 2.4e9 / 2) * c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
    else:
        assert r["physical_bound"] == "l1_tag_lookup_rate" and 0.0 < r["l2_miss_frac_of_hbm_peak"] <= 1.0
        want = c["TCP_TOTAL_CACHE_ACCESSES_sum"] * launches / (t * 2.4e9) / 256 / 1.4
        assert abs(r["l2_miss_frac_of_hbm_peak"] - dv["l2_miss_bytes_per_launch"] * launches / t / 8e12) < 1e-3
    assert abs(r["physical_frac"] - want) < 2e-3
    assert bench.physical_roofline(scene, "pruned", lds, kernel_ms, launches) == {}          # no PMC run of the pruned traversal


def test_parity_rows_compares_every_float():
    import torch
    rng = np.random.default_rng(0)
    frame = rng.random((48, 16, 3)).astype(np.float32)
    frame[3, 2, 1] = np.float32(-0.0)
    for k in (1, 6):
        rows = frame[::k].copy()
        ok = bench.parity_rows(torch.from_numpy(frame), rows, k)
        assert ok == {"rows": rows.shape[0], "row_stride": k, "bit_identical": True}
        rows[1, 5, 0] = np.nextafter(rows[1, 5, 0], np.float32(2.0))
        bad = bench.parity_rows(torch.from_numpy(frame), rows, k)
        assert bad["bit_identical"] is False and bad["differing_pixels"] == 1 and bad["max_abs_diff"] > 0
    rows = frame.copy()
    rows[3, 2, 1] = np.float32(0.0)                                    # +0 for -0: equal as numbers, not as bits
    assert bench.parity_rows(torch.from_numpy(frame), rows, 1)["bit_identical"] is False
