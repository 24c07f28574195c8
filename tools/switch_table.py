"""Every run-time switch of the product, in one table: docs/SWITCHES.md is GENERATED from this file and checked against the source by
tests/test_host_and_abi.py::test_switch_table_matches_the_source (a getenv("MOSFHET_...") the table does not know, or a row whose switch is gone, fails).

    python tools/switch_table.py            prints the table        python tools/switch_table.py --write      rewrites docs/SWITCHES.md

Columns: name; default; what it selects; the C-ABI setter that does the same at run time (if any); the test that covers more than one setting.
None of them changes a result -- every row selects between forms that are bit-identical by test -- with ONE exception, named in its row: MOSFHET_HIP_SPLIT_MAX.
"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name: (default, what it selects, run-time setter, covered by)
SWITCHES = {
    "MOSFHET_HIP_TEAM_MAX": ("512", "N = 1024: batches up to this size take the latency kernel (`pbs_team_kernel`, one workgroup of 2l wavefronts per ciphertext); 0 = never",
                             "`mosfhet_hip_set_team_max_batch`", "fixture `kernel_choice` (every bootstrap parity test runs with both kernels), `test_composition_batch_sizes`"),
    "MOSFHET_HIP_WIDE_TEAM_MAX": ("512", "N = 2048 (half the value at N = 4096): the same switch-over for `pbs_wide_team_kernel` / `pbs_wide_pair_kernel`; 0 = never",
                                  "`mosfhet_hip_set_wide_team_max_batch`", "`test_team_pacing_changes_timing_only`, `test_composition_batch_sizes`"),
    "MOSFHET_HIP_SPLIT_MAX": ("-1 (CUs / 2)", "N = 2048, l = 2, 4, 6: batches up to this size take TWO CUs per bootstrap (`pbs_split_kernel`: one workgroup per accumulator component, one "
                              "16 KiB exchange per CMUX step); 0 = never.  **The one switch that changes bits**: that kernel adds the rows of an external product per component "
                              "(FFT-level rounding apart from the other kernels' order; bit-identical to the oracle's by-component order)",
                              "`mosfhet_hip_set_split_max_batch`", "fixtures `kernel_choice` / `product_order`, tests marked `split_kernel`"),
    "MOSFHET_HIP_SPLIT_LIMIT": ("200000", "bound (10 ns ticks: 2 ms) of the wait of a pair's first workgroup for its partner, after which it takes the bootstrap alone (same bits); "
                                "0 = always alone (test switch)", "`mosfhet_hip_set_split_wait_limit`", "`test_split_kernel_batch_sizes_pairs_and_alone`"),
    "MOSFHET_HIP_WIDE_PAIRS": ("1", "N = 2048, even l, at most one ciphertext per CU: the latency kernel takes its rows two at a time (`pbs_wide_pair_kernel`); 0 = single rows",
                               "-", "`test_team_pacing_changes_timing_only`"),
    "MOSFHET_HIP_EP_PAIRS": ("1", "N = 2048, l = 4 external products: 0 forces the plain unit loop (`external_product_kernel` FORM 1) instead of the soaked pipelined one",
                             "`mosfhet_hip_set_ep_plain_loop`", "`test_external_product_loop_forms_and_the_scratch_guard`"),
    "MOSFHET_HIP_ROUND_CHUNK": ("4 x CUs at N = 2048", "bootstraps per launch when the key exceeds the L2s (one residency round per launch); 0 = one launch for the batch",
                                "-", "`tools/launch_chunk_sweep.py` (timing); results independent of it by construction (same kernel, same blocks)"),
    "MOSFHET_HIP_PACE": ("32", "N >= 2048: the teams of a residency round re-align per XCD every this many CMUX steps (`pace_teams`); 0 = off", "-",
                         "`test_team_pacing_changes_timing_only`"),
    "MOSFHET_HIP_PACE_SKIP": ("16", "paced launches of a device that skip the rendezvous after one launch's wait ran out (the chip is being shared); 0 = every launch tries", "-",
                              "`test_team_pacing_changes_timing_only`"),
    "MOSFHET_HIP_PACE_LIMIT": ("100000", "bound of one re-alignment wait in 10 ns ticks (1 ms), after which the launch stops waiting", "-", "`test_team_pacing_changes_timing_only`"),
    "MOSFHET_HIP_CB_TOGETHER": ("auto", "circuit bootstraps / KS21 on few inputs: all gadget levels through one table switch and one row-mode bootstrap launch (1), one per level (0); "
                                "auto = when it saves table sweeps, folded keys only", "-", "`test_per_level_compositions_on_the_lvl2_ring_in_row_mode`, `tools/cb_together_ab.sh`"),
    "MOSFHET_HIP_UNFOLD2_DFT": ("1", "unfolding-2 keys: per-group TRGSW assembled in the DFT domain (1) or in the torus domain like u = 4, 8 (0); read when a key is created",
                                "`mosfhet_hip_set_unfold2_dft`", "`test_functional_bootstrap_unfolded`"),
    "MOSFHET_HIP_UNFOLD_SPLIT_MAX": ("-1 (48 / 64)", "unfolded bootstraps: batches up to this size build all groups' selectors first (latency form); 0 = always the fused kernel",
                                     "`mosfhet_hip_set_unfold_split_max`", "`test_functional_bootstrap_unfolded`"),
    "MOSFHET_HIP_UNFOLD_BUDGET_GIB": ("2", "device memory the selectors-first form may take for its per-ciphertext selectors", "-", "`test_unfolded_bootstraps_full_size_lvl2`"),
    "MOSFHET_KS_SMALL_MAX": ("16", "table key switches: up to this many ciphertexts take the direct (row-gather) kernels instead of the tiled one; 0 = never", "-",
                             "`test_keyswitch_ragged_batches` (sizes on both sides of it)"),
    "MOSFHET_HIP_KS_WORDS": ("17", "table key switches with 2 - 4 digit bits: from this many ciphertexts on the word-lane kernel (`table_ks_words_kernel`: output words on the lanes, "
                             "wave-uniform digits, no LDS gather) instead of the ciphertext-lane tiles; 0 = never", "`mosfhet_hip_set_ks_words`",
                             "fixture `ks_form` (the key-switch parity tests run with both forms)"),
    "MOSFHET_HIP_NO_PEER": ("unset", "key replication between contexts: 1 = skip peer access, 2 = go through the pinned host buffer even on one device (test switch)", "-",
                            "`test_multi_device_compat`"),
    "MOSFHET_HIP_DEVICES": ("unset", "drop-in API: comma-separated device list, first = primary (same as `mosfhet_set_devices`)", "`mosfhet_set_devices`", "`test_multi_device_compat`"),
    "MOSFHET_HIP_DEVICE": ("0", "drop-in API: the one device to use when no list is given", "`mosfhet_set_devices`", "`tests/c/compat_suite.c`"),
    "MOSFHET_HIP_FULL_TABLE_KEYS": ("1", "drop-in API: table key-switch keys as full rows in HBM (1) or seed-compressed like the reference's default build (0: half the bytes, masks regenerated in the key switches; same results)", "-",
                                    "`test_seed_compressed_table_keys_are_the_same_keys`"),
    "MOSFHET_HIP_MARSHAL_THREADS": ("min(cores, 8)", "drop-in API: host threads that pack / unpack sample structs around the batched calls", "-", "`tests/c/compat_suite.c` case `big_batch`"),
    "MOSFHET_HIP_PIPE_CHUNK": ("1024", "drop-in API: ciphertexts per chunk of the two-stream upload / bootstrap / download pipeline of `*_batch`", "-", "`tools/compat_latency.c` (timing)"),
    "MOSFHET_HIP_PIPE_PIECE": ("256", "drop-in API: ciphertexts per download piece of the last chunk (unpacked while the next piece is in flight)", "-", "`tools/pipe_ab.sh` (timing)"),
    "MOSFHET_COMPAT_TLWE_PIECE": ("auto", "drop-in API: outputs per download piece of the generic TLWE batch path; 0 = one copy, then unpack", "-", "`tests/c/compat_suite.c`"),
}
# not library switches: the build's extra compiler flags and bench.py's test hooks (documented where they are read)
OTHER = {"MOSFHET_HIPCC_EXTRA": "mosfhet_amd/build.py: extra hipcc flags (part of the source hash)", "MOSFHET_BENCH_BACKEND": "bench.py test hook: process-group backend",
         "MOSFHET_BENCH_SHARE_GPU": "bench.py test hook: every rank on device 0", "MOSFHET_BENCH_FORCE_DIST": "bench.py test hook: process group at world size 1"}


def in_source():
    names = set()
    for path in glob.glob(os.path.join(ROOT, "mosfhet_amd", "csrc", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".hip", ".inc", ".h", ".c")):
            names.update(re.findall(r'getenv\("(MOSFHET_[A-Z0-9_]+)"\)', open(path).read()))
    return names


def markdown():
    lines = ["# Run-time switches of libmosfhet_hip.so", "",
             "GENERATED by `tools/switch_table.py --write` and held to the source by `tests/test_host_and_abi.py::test_switch_table_matches_the_source`.",
             "None of them changes a result -- each selects between forms that are bit-identical by the test named in the last column -- except `MOSFHET_HIP_SPLIT_MAX` "
             "(FFT-level different bits: see its row).", "",
             "| environment variable | default | selects | run-time setter | covered by |", "|---|---|---|---|---|"]
    for name in sorted(SWITCHES):
        lines.append("| `%s` | %s | %s | %s | %s |" % ((name,) + SWITCHES[name]))
    lines += ["", "Not library switches: " + "; ".join("`%s` (%s)" % kv for kv in sorted(OTHER.items())) + ".", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    if "--write" in sys.argv:
        os.makedirs(os.path.join(ROOT, "docs"), exist_ok=True)
        open(os.path.join(ROOT, "docs", "SWITCHES.md"), "w").write(markdown())
    else:
        print(markdown())
    missing, gone = in_source() - set(SWITCHES), set(SWITCHES) - in_source()
    if missing or gone:
        sys.exit("switch table out of date: not in the table %s, not in the source %s" % (sorted(missing), sorted(gone)))
