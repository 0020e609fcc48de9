"""HBM traffic of the GEMM kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; rocpd databases).
Usage: python tools/pmc_traffic.py <fetch.db> <write.db> <forwards> <gemm_launches_per_forward> > r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md,
HBM section), hence the correction factor."""
import json
import sqlite3
import sys

FAMILY = ("gemm_glds_kernel", "gemm_pp_kernel", "gemm_reg_kernel", "stem_conv_kernel", "head_tail_kernel")


def total(db, counter):
    cur = sqlite3.connect(db).cursor()
    fam = allk = 0.0
    for name, cname, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
        if cname != counter:
            continue
        allk += val
        if any(f in name for f in FAMILY):
            fam += val
    return fam, allk


def main():
    fetch_db, write_db, forwards, launches = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    f_fam, f_all = total(fetch_db, "FETCH_SIZE")
    w_fam, w_all = total(write_db, "WRITE_SIZE")
    corr = 2.0
    fam_bytes = (corr * f_fam + w_fam) * 1024.0 / forwards
    all_bytes = (corr * f_all + w_all) * 1024.0 / forwards
    print(json.dumps({
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, DPTX_STREAMS=1, bench.py B=32 bf16",
        "forwards": forwards,
        "gemm_family_fetch_kb_raw_per_forward": f_fam / forwards,
        "gemm_family_write_kb_per_forward": w_fam / forwards,
        "gfx950_fetch_correction": corr,
        "gemm_family_hbm_bytes_per_forward": fam_bytes,
        "all_kernels_hbm_bytes_per_forward": all_bytes,
        "gemm_launches_per_forward": launches,
        "gemm_family_hbm_bytes_per_launch": fam_bytes / launches,
    }, indent=1))


if __name__ == "__main__":
    main()
