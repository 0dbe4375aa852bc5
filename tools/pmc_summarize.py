#!/usr/bin/env python3
"""profiles/<tag>_pmc/*.csv (rows of rocprofv3 --pmc passes, tools/pmc_passes.sh) -> traffic.json, mfma_util.json.
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half of what a wide coalesced stream
fetches (MI355X_MICROARCH.md section HBM); an upper bound for narrow accesses.  Every record carries the sha1 of the
kernel sources it was measured on: benchmarks/workloads.py drops a record whose hash no longer matches the tree."""
import collections, csv, glob, hashlib, json, os, sys

dst = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(f):
    return hashlib.sha1(open(os.path.join(root, "pdanet_amd", "csrc", f), "rb").read()).hexdigest()[:16]


def rows(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def per_call(path, counter, calls, prefix="pda::"):
    """sum of `counter` over all pda:: kernels of the file / number of operator calls (a call may be several kernels)."""
    r = [x for x in rows(path) if x["Counter_Name"] == counter and prefix in x["Kernel_Name"]]
    if not r:
        return None, {}
    by = collections.Counter()
    for x in r:
        by[x["Kernel_Name"].split("(")[0]] += float(x["Counter_Value"])
    return sum(by.values()) / calls, {k: v / calls for k, v in by.items()}


CALLS = 5
traffic = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 tools/pmc_targets.py "
                   "<target> 5; raw pda:: rows next to this file.  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per operator call "
                   "(gfx950: FETCH_SIZE counts half of wide coalesced reads).",
           "csrc_sha1": {f: sha(f) for f in ("fps.hip", "ball_query.hip", "ball_query_cells.hip", "wgrad.hip")}, "kernels": {}}
for target, key in (("fps", "pda::fps_chain_kernel FPS 16384->4096 b2"), ("ball_query", "pda::ball_query 16384x16384 r2 b2"),
                    ("wgrad", "pda::linear_wgrad dW(512x512) over 131072 tokens")):
    f, fk = per_call(os.path.join(dst, "fetch_size_%s.csv" % target), "FETCH_SIZE", CALLS)
    w, wk = per_call(os.path.join(dst, "write_size_%s.csv" % target), "WRITE_SIZE", CALLS)
    if f is None or w is None:
        continue
    traffic["kernels"][key] = {"FETCH_SIZE_KB_per_call": f, "WRITE_SIZE_KB_per_call": w, "calls": CALLS,
                               "hbm_bytes_per_launch_mean": (2 * f + w) * 1024,
                               "by_kernel_KB": {k: {"FETCH_SIZE": fk.get(k, 0.0), "WRITE_SIZE": wk.get(k, 0.0)} for k in sorted(set(fk) | set(wk))}}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

util = {"_how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/pmc_targets.py <target> 5.  "
                "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): busy SIMD-cycles over kernel cycles x 1024 SIMDs "
                "(GRBM_GUI_ACTIVE is summed over the 8 XCDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel duration.",
        "csrc_sha1": {f: sha(f) for f in ("wgrad.hip", "sa_mlp.hip", "gemm_split.hip", "sa_train_small.hip")}, "kernels": {}}
for target in ("wgrad", "sa_mlp", "lin_cols", "lin_split", "gemm_split", "sa_small"):
    agg = collections.OrderedDict()
    for x in rows(os.path.join(dst, "mfma_busy_%s.csv" % target)):
        name = x["Kernel_Name"]
        if not any(k in name for k in ("wgrad_kernel", "wgrad_split_kernel", "sa_mlp_kernel", "lin_cols_kernel", "lin_split_kernel", "gemm_split_wide_kernel", "gemm_split_kernel", "ss_fwd_kernel", "ss_bwd_kernel")):
            continue
        key = "%s grid=%s" % (name.split("(")[0].replace("void ", ""), x["Grid_Size"])
        d = agg.setdefault(key, {"n": collections.Counter(), "v": collections.Counter(), "ns": 0.0})
        d["n"][x["Counter_Name"]] += 1; d["v"][x["Counter_Name"]] += float(x["Counter_Value"])
        if x["Counter_Name"] == "GRBM_GUI_ACTIVE":
            d["ns"] += float(x["End_Timestamp"]) - float(x["Start_Timestamp"])
    for key, d in agg.items():
        n = d["n"]["GRBM_GUI_ACTIVE"]
        if not n or not d["n"]["SQ_VALU_MFMA_BUSY_CYCLES"]:
            continue
        busy, act, ns = d["v"]["SQ_VALU_MFMA_BUSY_CYCLES"] / n, d["v"]["GRBM_GUI_ACTIVE"] / n, d["ns"] / n
        util["kernels"][key] = {"launches": n, "avg_ns": ns, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": act,
                                "mfma_util": busy / (act / 8 * 1024), "clock_ghz": act / 8 / ns}
json.dump(util, open(os.path.join(dst, "mfma_util.json"), "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch_mean"]) for k, v in traffic["kernels"].items()}, indent=1))
print(json.dumps({k: round(v["mfma_util"], 4) for k, v in util["kernels"].items()}, indent=1))
