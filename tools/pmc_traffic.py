#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel and per-class HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01h

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KiB; FETCH_SIZE
tallies a wide coalesced 128-byte read request as 64 bytes, so fetched bytes = 2 x FETCH_SIZE KiB; WRITE_SIZE is exact.
"""
import csv, glob, json, os, sys
from collections import defaultdict

CLASSES = {   # bench.py --roofline-kernel name -> predicate on the kernel name
    "gemm": lambda n: ("gemm_nt_kernel" in n or "gemm_ws_kernel" in n or "gemm_ws_lnbwd_kernel" in n) and not is_ln(n),
    "gemm_ln": lambda n: ("gemm_nt_kernel" in n or "gemm_ws_kernel" in n or "gemm_kl_kernel" in n) and is_ln(n),
    # what bench.py's wgrad class brackets: one scope per product = wgrad_ks_kernel + its reduce pass, or one wgrad_kernel launch
    # (NOT tail_wgrad_kernel, which "wgrad_kernel" in n would also match).  Per LAUNCH of the class = per product: the reduce
    # pass's bytes are added to its product's, so the class count is the number of wgrad_ks + wgrad_kernel dispatches.
    "wgrad": lambda n: ("wgrad_ks_kernel" in n or "wgrad_ks_reduce" in n or "ge2e::wgrad_kernel" in n or "ge2e12wgrad_kernel" in n or "prenet_bwd_kernel" in n),
    "wgrad_reduce": lambda n: "wgrad_ks_reduce" in n,
    "ffn": lambda n: "ffn_chain_kernel" in n,
    "attn_fwd": lambda n: "attn_fwd_kernel" in n,
    "attn_bwd": lambda n: "attn_bwd_kernel" in n,
}


def is_ln(n):
    # EPI_LN == 3: gemm_nt_kernel<T,64,256,32,128,3,...>, gemm_ws_kernel<3,256>, gemm_kl_kernel<3,...>
    # (round 2: the streaming kernels are templated on the storage type first: gemm_ws_kernelIDF16bLi3E... / IDF16_Li3E...)
    return ("Li64ELi256ELi32ELi128ELi3E" in n or "gemm_ws_kernelILi3E" in n or "gemm_ws_kernel<3," in n
            or "gemm_kl_kernelILi3E" in n or "gemm_kl_kernel<3," in n
            or "gemm_ws_kernelIDF16bLi3E" in n or "gemm_ws_kernelIDF16_Li3E" in n or "gemm_kl_kernelIDF16" in n)


def read_pass(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    per = defaultdict(lambda: [0.0, set()])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            per[k][0] += float(r["Counter_Value"])
            per[k][1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in per.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = read_pass(fdir, "FETCH_SIZE"), read_pass(wdir, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        if "ge2e" not in k:
            continue
        fk, fn = fetch.get(k, (0.0, 0))
        wk, wn = write.get(k, (0.0, 0))
        n = max(fn, wn, 1)
        rows.append((k, n, fk / max(fn, 1), 2.0 * fk * 1024 / max(fn, 1), wk / max(wn, 1), wk * 1024 / max(wn, 1)))
    with open(out + "_pmc_hbm_traffic_per_kernel.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch_raw", "fetch_bytes_per_launch_corrected_x2",
                    "WRITE_SIZE_KB_per_launch", "write_bytes_per_launch"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.0f}", f"{r[4]:.1f}", f"{r[5]:.0f}"])
    classes = {}
    for name, pred in CLASSES.items():
        tot, n = 0.0, 0
        for r in rows:
            if pred(r[0]):
                tot += (r[3] + r[5]) * r[1]
                if not (name == "wgrad" and "wgrad_ks_reduce" in r[0]):     # a reduce pass belongs to its products' launches
                    n += r[1]
        if n:
            classes[name] = round(tot / n)
    json.dump(classes, open(out + "_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(classes))


if __name__ == "__main__":
    main()
