"""k_scan_reduce (round 6, opt-in PCR_PHASE_SPLIT=1): search and reduce of a mid-size scan over a point target in ONE launch,
phase-split (csrc/kernels.hip).  Measured slower than the kernel pair at every pose (profiles/r06_phase_split_null.txt), so it
is not the default; it stays exact: matches bit for bit those of the kernel pair, sums within 1e-12 of the pair's (another
association of the same float64 terms) and 1e-9 of the oracle's, device-resident loop == host loop bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

CODE = r"""
import json, sys, numpy as np
sys.path.insert(0, %r)
from point_cloud_registration_amd import _capi as capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan, street_normals
ctx = capi.get_context(0)
target = street(400_000, seed=41); scan, T_true = perturbed_scan(target, None, seed=42)
tgt = capi.Target.points(ctx, target, street_normals(target)); sc = capi.Scan(ctx, scan)
Tm = np.eye(4); Tm[:3, 3] = [0.02, -0.01, 0.03]
out = {}
for name, kind in (("icp", capi.ICP), ("plane", capi.PLANE)):
    rows = []
    for T in (np.eye(4), Tm, T_true):
        o = capi.linearize(tgt, sc, kind, T, 2.0)
        rows.append({"o": [float(v) for v in o], "m": int(np.asarray(sc.matches(), dtype=np.int64).sum()), "c": int((np.asarray(sc.matches()) >= 0).sum())})
    Td, itd = capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, flags=capi.FLAG_ICP_RR_QUIRK | capi.FLAG_DEVICE_LOOP)
    Th, ith = capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0, flags=capi.FLAG_ICP_RR_QUIRK | capi.FLAG_HOST_LOOP)
    out[name] = {"rows": rows, "itd": int(itd), "ith": int(ith), "same": bool(np.array_equal(Td, Th)), "T": [float(v) for v in np.asarray(Td).ravel()]}
ctx.profile_enable(True); ctx.profile_reset(); capi.linearize(tgt, sc, capi.PLANE, np.eye(4), 2.0); prof = ctx.profile_read(); ctx.profile_enable(False)
out["kernels"] = {k: int(v[0]) for k, v in prof.items()}
print("RESULT " + json.dumps(out))
""" % REPO


def _run(split):
    env = dict(os.environ, PCR_PHASE_SPLIT=str(split))
    env.pop("PCR_LIB", None)
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_phase_split_kernel_is_exact():
    from oracle import oracle as orc
    from point_cloud_registration_amd.synthetic import street, perturbed_scan, street_normals
    pair, split = _run(0), _run(1)
    # the launches really differ: one "linearize" launch against search + reduce
    assert pair["kernels"]["nn"] == 1 and pair["kernels"]["reduce"] == 1 and pair["kernels"]["linearize"] == 0
    assert split["kernels"]["linearize"] == 1 and split["kernels"]["nn"] == 0 and split["kernels"]["reduce"] == 0
    target = street(400_000, seed=41); scan, T_true = perturbed_scan(target, None, seed=42)
    ot = orc.TargetPoints(target, normals=street_normals(target))
    Tm = np.eye(4); Tm[:3, 3] = [0.02, -0.01, 0.03]
    for name, okind in (("icp", orc.ICP), ("plane", orc.PLANE)):
        for k, T in enumerate((np.eye(4), Tm, T_true)):
            a, b = pair[name]["rows"][k], split[name]["rows"][k]
            assert a["m"] == b["m"] and a["c"] == b["c"], (name, k)            # the same matches
            oa, ob = np.array(a["o"]), np.array(b["o"])
            assert np.max(np.abs(oa - ob)) <= 1e-12 * np.max(np.abs(oa)), (name, k)
            if k != 1:
                Ho, go, e2o = orc.calc_H_g_e2(okind, ot, T, scan, 2.0)
                H = np.zeros((6, 6)); H[np.triu_indices(6)] = ob[:21]; H = H + np.triu(H, 1).T
                assert np.max(np.abs(H - Ho)) <= 1e-9 * np.max(np.abs(Ho)), (name, k)
                assert abs(ob[27] - e2o) <= 1e-9 * max(abs(e2o), 1.0)
        assert split[name]["same"] and split[name]["itd"] == split[name]["ith"] == pair[name]["itd"]
        assert np.max(np.abs(np.array(split[name]["T"]) - np.array(pair[name]["T"]))) < 1e-9
