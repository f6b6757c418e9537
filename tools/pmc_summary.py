#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_all.sh into the two small files that are committed under profiles/:
<tag>_pmc_summary.json (per-launch counter means of taco_step_kernel, HBM bytes per env-step with the gfx950 corrections of
MI355X_MICROARCH.md) and <tag>_kernel_stats_bench_4096.csv (the --stats kernel table of the bench run)."""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
O = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
DROP = 3            # short passes (the one-wavefront instruction count at 4 096 envs): reset-all launch + warm-up
STEADY_DROP = 500   # long passes (>= 600 launches): everything before the steady state -- 1.1 % of the envs reset in every step from ~ step 300 on, and a
                    # resetting env costs its wavefront instructions and stores that the first dozen launches of a fresh env never show (round 5: 26.8 M
                    # VALU instructions per launch at 262 144 envs in launches 3..11, 28.0 M in the steady state)


def counters(d):
    out = {}
    for f in glob.glob(os.path.join(O, d, "*", "*counter_collection.csv")):
        per = {}
        for r in csv.DictReader(open(f)):
            if "taco_step_kernel" not in r["Kernel_Name"]:
                continue
            per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for name, by in per.items():
            vals = [by[k] for k in sorted(by)]
            vals = vals[STEADY_DROP if len(vals) >= 600 else DROP:]
            out[name] = {"launches_averaged": len(vals), "mean_per_launch": sum(vals) / max(len(vals), 1)}
    return out


S = {"sq_262144": counters(f"{tag}_pmc_sq_262144"), "sq_4096": counters(f"{tag}_pmc_sq_4096"), "sq_4096_quad": counters(f"{tag}_pmc_sq_4096_quad")}
for n in (262144, 4096):
    S[f"fetch_{n}"] = counters(f"{tag}_pmc_FETCH_SIZE_{n}")
    S[f"write_{n}"] = counters(f"{tag}_pmc_WRITE_SIZE_{n}")
der = {"note": "FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of "
               "16-B-per-lane coalesced reads); separate --pmc passes; launches 500..699 of a fresh env (the steady state: 1.1 % of the envs reset per step) "
               "except per_wave_4096_quad (launches 3..39: the reset-free instruction count of one step wavefront)"}
for n in (262144, 4096):
    try:
        rd = S[f"fetch_{n}"]["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
        wr = S[f"write_{n}"]["WRITE_SIZE"]["mean_per_launch"] * 1024
        der[f"hbm_bytes_per_launch_{n}"] = rd + wr
        der[f"hbm_bytes_per_env_step_{n}"] = (rd + wr) / n
        der[f"read_B_per_env_step_{n}"] = rd / n
        der[f"write_B_per_env_step_{n}"] = wr / n
    except KeyError:
        pass
sq = S["sq_262144"]
if "SQ_WAVES" in sq and sq["SQ_WAVES"]["mean_per_launch"]:
    w = sq["SQ_WAVES"]["mean_per_launch"]
    der["per_wave_262144"] = {k: v["mean_per_launch"] / w for k, v in sq.items() if k != "SQ_WAVES"}
sq4 = S["sq_4096_quad"]
if "SQ_WAVES" in sq4 and sq4["SQ_WAVES"]["mean_per_launch"]:
    w = sq4["SQ_WAVES"]["mean_per_launch"]
    der["per_wave_4096_quad"] = {k: v["mean_per_launch"] / w for k, v in sq4.items() if k != "SQ_WAVES"}
    der["instr_per_step_wavefront_4096"] = sum(sq4[k]["mean_per_launch"] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH") if k in sq4) / w
S["derived"] = der
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taco_amd import build as _build  # noqa: E402
S["source_hash"] = _build.source_hash()  # bench.py reports roofline.traffic only for the build it was measured on
json.dump(S, open(os.path.join(O, f"{tag}_pmc_summary.json"), "w"), indent=1)
for f in glob.glob(os.path.join(O, f"{tag}_stats", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(O, f"{tag}_kernel_stats_bench_4096.csv"))
print(json.dumps(der, indent=1))
