"""Rewrites the r<NN> entries of profiles/roofline_traffic.json (what bench.py quotes as roofline.traffic and
distance_matrix_build.rocprof_mean_us / write_size_bytes) from the summaries tools/profile_bench.sh and tools/profile_r03.sh
left in profiles/.  usage: update_traffic.py [tag]   (default r03)"""
import csv, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
BEST = "void k_cluster_two_opt<6, true, 1, HIP_vector_type<float, 2u>, true, false>"


def pmc(name, kernel, counter):
    return json.load(open(os.path.join(P, "%s_%s.json" % (tag, name))))[kernel][counter]["mean"]


def mean_us(name, kernel):
    for row in csv.DictReader(open(os.path.join(P, "%s_%s.csv" % (tag, name)))):
        if row["kernel"] == kernel: return float(row["mean_us"]), int(row["calls"])
    raise KeyError(kernel)


path = os.path.join(P, "roofline_traffic.json")
d = json.load(open(path))
if tag >= "r04":
    # round 4: bench.py's timed kernels are the exhaustive sweep's (k_move_pos + k_exh, one pair per sweep)
    KE, KM = "void k_exh<6, true, 4>", "void k_move_pos<6, true>"
    fe, we = pmc("pmc_FETCH_SIZE", KE, "FETCH_SIZE"), pmc("pmc_WRITE_SIZE", KE, "WRITE_SIZE")
    fm, wm = pmc("pmc_FETCH_SIZE", KM, "FETCH_SIZE"), pmc("pmc_WRITE_SIZE", KM, "WRITE_SIZE")
    total = int(round((2 * fe + we + 2 * fm + wm) * 1024))
    d["%s_exhaustive_sweep_n10000_hbm_bytes_per_launch" % tag] = total
    d["%s_derivation" % tag] = {
        "kernels": {"tsp::k_exh<6 (EUC_2D integer-coordinate variant), true, RJ = 4>": {"FETCH_SIZE_KiB_mean": fe, "WRITE_SIZE_KiB_mean": we},
                    "tsp::k_move_pos<6, true>": {"FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm}},
        "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 per kernel, summed: one launch pair = one sweep = 49 985 000 delta expressions (raw sum = %d bytes)" % int(round((fe + we + fm + wm) * 1024)),
        "algorithmic_operand_bytes_per_launch": 72 * 49985000,
        "note": "the sweep's operands stay on chip: k_move_pos rewrites the 0.3 MB of position-ordered arrays, k_exh reads them once per wave (columns) and through the scalar cache (rows); HBM traffic is 0.1 % of SURVEY 8(d)'s 72 B per delta -- the kernel is VALU-bound",
        "source": ["profiles/%s_pmc_FETCH_SIZE.json" % tag, "profiles/%s_pmc_WRITE_SIZE.json" % tag, "profiles/%s_kernel_stats.csv" % tag]}
    json.dump(d, open(path, "w"), indent=1)
    print(tag, "exhaustive sweep bytes per launch", total)
    sys.exit(0)
f, w = pmc("pmc_FETCH_SIZE", BEST, "FETCH_SIZE"), pmc("pmc_WRITE_SIZE", BEST, "WRITE_SIZE")
sweeps = 1428
total = int(round((2 * f + w) * 1024))
d["%s_cluster_descent_n10000_hbm_bytes_per_launch" % tag] = total
d["%s_derivation" % tag] = {
    "kernel": "tsp::k_cluster_two_opt<6 (EUC_2D integer-coordinate variant), true, BEST, float2, sorted> -- one launch = one descent of %d sweeps" % sweeps,
    "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w,
    "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md HBM section; the reads are the 8-byte sc1 polls of the candidate exchange and the one-time load of 256 replicas, for which the factor 2 is not calibrated: raw sum = %d bytes)" % int(round((f + w) * 1024)),
    "per_sweep_bytes": total // sweeps,
    "source": ["profiles/%s_pmc_FETCH_SIZE.json" % tag, "profiles/%s_pmc_WRITE_SIZE.json" % tag],
    "note": "the memory-side traffic of a descent is the candidate exchange (sc1 stores and polls bypass the caches by design: 100 % non-algorithmic bytes, harmless for bandwidth -- it is the latency term); every candidate is written to eight copies of the exchange area (one per XCD: csrc/two_opt_cluster.hip, cl_exchange), hence eight times the store bytes of round 2; the tour is read once per launch",
}
dm = {}
for key, kern, alg in (("int32", "void k_dist_matrix<6, true, int>", 4e8), ("f64", "void k_dist_matrix<6, true, double>", 8e8)):
    us, calls = mean_us("kernel_stats_dist_matrix", kern)
    wb = int(round(pmc("pmc_WRITE_SIZE_dist_matrix", kern, "WRITE_SIZE") * 1024))
    dm[key] = {"rocprof_mean_us": us, "write_size_bytes": wb, "algorithmic_bytes": int(alg), "hbm_GBps_write_size_over_rocprof_time": wb / us / 1e3}
dm["source"] = ["profiles/%s_kernel_stats_dist_matrix.csv (rocprofv3 --kernel-trace --stats of tools/dist_matrix_time.py: %d launches each, rotating over 3 output buffers)" % (tag, calls),
                "profiles/%s_pmc_WRITE_SIZE_dist_matrix.json (separate --pmc WRITE_SIZE pass; KiB per dispatch, exact for 16-byte streaming stores)" % tag]
d["%s_dist_matrix" % tag] = dm
json.dump(d, open(path, "w"), indent=1)
print(tag, "cluster descent bytes per launch", total, "| dist matrix", {k: (v["rocprof_mean_us"], round(v["hbm_GBps_write_size_over_rocprof_time"])) for k, v in dm.items() if k != "source"})
