"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the headers declare, the
MOSFHET-compatible host layer (key / sample generation, phases, packing) is correct, and compute calls fail
loudly without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


@pytest.mark.parametrize("header", ["mosfhet_hip.h", "mosfhet_compat.h"])
def test_library_exports_every_declared_symbol(native_lib, header):
    names = declared_functions(header)
    assert len(names) > 15, names
    missing = [n for n in names if not hasattr(native_lib, n)]
    assert not missing, "declared in include/%s but not exported by libmosfhet_hip.so: %s" % (header, missing)


# the signatures a program written against the reference's mosfhet.h for this path links to (reference include/mosfhet.h lines
# 179-182, 190, 263-264, 296, 342-344, 374-457), plus what the reference's own leveled-LUT application needs around them
MUST_KEEP = """new_bootstrap_key free_bootstrap_key blind_rotate functional_bootstrap_wo_extract functional_bootstrap programmable_bootstrap
trgsw_mul_trlwe_DFT polynomial_torus_to_DFT polynomial_DFT_to_torus polynomial_mul_DFT polynomial_mul_addto_DFT trlwe_from_DFT trlwe_to_DFT trgsw_to_DFT
trlwe_extract_tlwe tlwe_new_KS_key tlwe_keyswitch init_fft blind_rotate_ga trlwe_eval_automorphism public_mux
multivalue_bootstrap_CLOT21 multivalue_bootstrap_phase1 multivalue_bootstrap_phase2 circuit_bootstrap circuit_bootstrap_2 circuit_bootstrap_3
full_domain_functional_bootstrap full_domain_functional_bootstrap_KS21 full_domain_functional_bootstrap_KS21_2 full_domain_functional_bootstrap_CLOT21
full_domain_functional_bootstrap_CLOT21_2 functional_bootstrap_trgsw_phase1 functional_bootstrap_trgsw_phase2 new_bootstrap_key_ga
functional_bootstrap_ga functional_bootstrap_wo_extract_ga free_bootstrap_key_ga trlwe_keyswitch trlwe_priv_keyswitch trlwe_priv_keyswitch_2
trlwe_packing1_keyswitch trlwe_tensor_prod_FFT tlwe_mul trlwe_new_RL_key polynomial_permute inverse_mod_2N
tlwe_alloc_sample trlwe_alloc_new_sample trgsw_alloc_new_sample trlwe_alloc_new_DFT_sample trgsw_alloc_new_DFT_sample trgsw_alloc_new_DFT_sample_array
polynomial_new_DFT_polynomial tlwe_new_binary_key trlwe_new_binary_key trgsw_new_key tlwe_sample trlwe_sample trgsw_monomial_sample tlwe_phase trlwe_phase
trlwe_torus_packing free_tlwe free_trlwe free_trgsw free_polynomial safe_malloc generate_random_bytes""".split()


def test_must_keep_symbols_are_exported(native_lib):
    """nm-level check of SURVEY 8(b)'s must-keep signatures: every name is a defined dynamic symbol of libmosfhet_hip.so and is declared by
    include/mosfhet.h (through mosfhet_compat.h)."""
    import subprocess
    from mosfhet_amd import engine
    out = subprocess.run(["nm", "-D", "--defined-only", engine.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    missing = [n for n in MUST_KEEP if n not in exported]
    assert not missing, "not exported: %s" % missing
    declared = set(declared_functions("mosfhet_compat.h"))
    undeclared = [n for n in MUST_KEEP if n not in declared]
    assert not undeclared, "exported but not declared in include/mosfhet_compat.h: %s" % undeclared
    assert '#include "mosfhet_compat.h"' in open(os.path.join(ROOT, "include", "mosfhet.h")).read()


NOT_PROVIDED = """""".split()


def _prototypes(path):
    """name -> (return type, [argument types]) of every function a C header declares, parameter names and spacing removed"""
    import re
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    scalars = {"int", "double", "void", "uint64_t", "bool", "char", "uint8_t", "uint16_t", "int64_t", "uint32_t", "FILE", "size_t"}
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{()]*)\)\s*;", text):
        ret, name, args = re.sub(r"\s+", "", m.group(1)).replace("const", ""), m.group(2), m.group(3).strip()
        types = []
        for a in ([] if args in ("", "void") else args.split(",")):
            toks = re.sub(r"\[\d*\]", "*", a).replace("*", " * ").split()
            if len(toks) > 1 and re.match(r"^[A-Za-z_]\w*$", toks[-1]) and toks[-1] not in scalars and not toks[-1][0].isupper():
                toks = toks[:-1]          # the parameter's name
            types.append("".join(toks).replace("const", ""))
        out[name] = (ret, types)
    return out


@pytest.mark.skipif(not os.path.exists("/root/reference/include/mosfhet.h"), reason="the reference tree exists in the build container only")
def test_header_matches_the_reference_prototypes():
    """Drop-in means: what include/mosfhet.h declares has the reference's prototypes, argument by argument, and what it does not declare is the
    list INTEGRATION.md gives (anything else missing is a regression)."""
    ref = _prototypes("/root/reference/include/mosfhet.h")
    ours = _prototypes(os.path.join(ROOT, "include", "mosfhet_compat.h"))
    common = sorted(set(ref) & set(ours))
    different = [(n, ref[n], ours[n]) for n in common if ref[n] != ours[n]]
    assert not different, different[:5]
    missing = sorted(set(ref) - set(ours))
    assert missing == sorted(NOT_PROVIDED), (sorted(set(missing) - set(NOT_PROVIDED)), sorted(set(NOT_PROVIDED) - set(missing)))
    assert len(common) == 275


@pytest.mark.skipif(not os.path.isdir("/root/reference/applications"), reason="the reference tree exists in the build container only")
def test_reference_application_relinks_unchanged(native_lib):
    """The reference's own applications/leveled_lut/vertical_packing.c compiles UNCHANGED against include/mosfhet.h and links to the product library
    (oracle/ref/Makefile target `app`; the binary runs under -m gpu: test_reference_application_runs_on_the_gpu)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle", "ref"), "app"])
    exe = os.path.join(ROOT, "oracle", "_ref", "vertical_packing_hip")
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True, check=True).stdout
    assert "libmosfhet_hip.so" in ldd and "not found" not in ldd, ldd
    undefined = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    for name in ("trgsw_mul_trlwe_DFT", "trlwe_from_DFT", "trgsw_to_DFT", "blind_rotate", "trgsw_alloc_new_DFT_sample_array"):
        assert name in undefined, "the application does not import %s?" % name
    # the same for its leveled-LUT example and its two benchmark programs (test/benchmark.c is the reference's headline benchmark)
    exported = set(line.split()[-1] for line in
                   subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "mosfhet_amd", "libmosfhet_hip.so")], capture_output=True, text=True, check=True).stdout.splitlines() if line.strip())
    for prog, needs in (("leveled_lut_hip", ("trgsw_monomial_sample", "trgsw_mul_trlwe_DFT")), ("benchmark_arith_hip", ("polynomial_naive_mul_addto_torus", "polynomial_add_DFT_polynomials")),
                        ("benchmark_hip", ("multivalue_bootstrap_UBR_phase2", "functional_bootstrap_trgsw_phase1", "new_bootstrap_key_ga")),
                        ("ufhe_tests_hip", ("trlwe_packing_keyswitch", "trlwe_new_packing_KS_key", "tlwe_keyswitch", "functional_bootstrap", "multivalue_bootstrap_phase1"))):
        exe = os.path.join(ROOT, "oracle", "_ref", prog)
        ldd = subprocess.run(["ldd", exe], capture_output=True, text=True, check=True).stdout
        assert "libmosfhet_hip.so" in ldd and "not found" not in ldd, (prog, ldd)
        und = [line.split()[-1] for line in subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout.splitlines() if line.strip()]
        for name in needs:
            assert name in und and name in exported, (prog, name)


def test_host_helpers_are_clean_under_sanitizers(native_lib, tmp_path):
    """The host layer's sources compiled with -fsanitize=address,undefined (+ leak check) into a program that calls every host-only helper once
    (tests/c/host_helpers.c): no report, every allocation freed."""
    import subprocess
    host = os.path.join(ROOT, "mosfhet_amd", "csrc", "host")
    srcs = [os.path.join(host, f) for f in ("mosfhet_compat.c", "mosfhet_compat_dft.c", "mosfhet_compat_multi.c", "mosfhet_compat_legacy.c", "mosfhet_compat_extra.c", "csprng.c")]
    exe = str(tmp_path / "host_helpers")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=shift", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "mosfhet_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "host_helpers.c")] + srcs +
                          ["-o", exe, "-L" + os.path.join(ROOT, "mosfhet_amd"), "-lmosfhet_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "mosfhet_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "host helpers ok" in r.stdout and "ERROR" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-3000:]
    assert "exponent of TRGSW(X^5): 5" in r.stdout


def test_kernel_table_of_the_build(native_lib):
    """The built library's own kernel table (tools/kernel_table.py: code-object metadata, no GPU).  What the launcher ASSUMES of the build is held here, so that a
    toolchain change shows up on the CPU box and not as a silent change of the kernels that run:
      * the software-pipelined unit loop on two-wavefront teams (external_product_kernel<Fft2048L, 4, *, no CMUX>, FORM 0) is taken only while that build has no
        scratch (capi.hip: ep_go falls back to the plain loop otherwise: correct, 11 % slower at lvl2) -- today it has none;
      * its fall-back (FORM 1) exists for exactly those instantiations."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_table
    rows = kernel_table.table()
    by_name = {r["name"]: r for r in rows}
    piped = [r for r in rows if r["name"].startswith("external_product_kernel<Fft2048T<false, true>, 4, ") and r["name"].endswith(", false, 0>")]
    assert len(piped) == 2, [r["name"] for r in piped]
    for r in piped:
        assert r["scratch"] == 0, "%s now spills %d bytes: the launcher will run the plain loop (re-measure, re-soak: tools/soak.py)" % (r["name"], r["scratch"])
        assert r["name"][:-2] + "1>" in by_name, "no plain-loop fall-back for " + r["name"]
    assert len(rows) < 330, "%d kernel instantiations" % len(rows)


def test_no_lds_reads_emitted_behind_a_workgroup_barrier():
    """The root cause of the "wrong units" of rounds 3 - 5 (experiments/README.md, Round 5) was the COMPILER moving LDS reads that the source places in front of a
    workgroup barrier behind it (machine sinking into the block after a conditional branch; S_BARRIER is no store to that pass).  Every barrier of the library now goes
    through workgroup_sync() (negacyclic_fft.h).  This holds the generated code to it: the external-product kernel in the loop forms and rings that used to fail, and
    lvl2's production form, are compiled to device listings (seconds, no GPU) and checked by tools/check_lds_barriers.py; the checker itself is held to a listing
    fragment with the old symptom."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_lds_barriers as chk
    bad_listing = "\n".join(["_Z1kv:", "\tds_write_b128 v1, v[2:5]", "\t; wave barrier", "\ts_waitcnt lgkmcnt(0)", "\ts_barrier", "\ts_cbranch_vccnz .LBB0_2", ".LBB0_2:",
                             "\tds_read_b128 v[2:5], v6", "\ts_endpgm"])
    good_listing = bad_listing.replace("\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n", "").replace("\tds_read_b128 v[2:5], v6\n", "\tds_read_b128 v[2:5], v6\n\ts_barrier\n")
    assert len(chk.check(bad_listing)) == 1 and chk.check(good_listing) == []
    # the two-wavefront exchanges (ds_write ; s_barrier ; ds_read ; s_barrier -- ADVICE round 5): reads moved behind the barrier that frees the buffer leave LDS writes followed
    # by two barriers with no read in between
    cross_bad = "\n".join(["_Z1kv:", "\tds_write_b128 v1, v[2:5]", "\ts_barrier", "\ts_barrier", "\ts_cbranch_scc1 .LBB0_3", ".LBB0_3:", "\tds_read_b128 v[2:5], v1", "\ts_endpgm"])
    cross_good = "\n".join(["_Z1kv:", "\tds_write_b128 v1, v[2:5]", "\ts_barrier", "\tds_read_b128 v[2:5], v1", "\ts_barrier", "\ts_barrier", "\ts_endpgm"])
    assert len(chk.check(cross_bad)) == 1 and chk.check(cross_good) == []
    assert chk.build_and_check() == []
    # the minimal reproducer (tools/spill_hazard/minimal_sink_past_barrier.hip): the form with workgroup_sync()'s clobbers must be clean; whether the raw form still shows
    # the reordering is a property of the toolchain and is only reported (today, ROCm 7.2.0: it does)
    import subprocess
    import tempfile
    import warnings
    with tempfile.TemporaryDirectory() as tmp:
        lst = os.path.join(tmp, "min.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-o", lst, os.path.join(ROOT, "tools", "spill_hazard", "minimal_sink_past_barrier.hip")],
                              stderr=subprocess.DEVNULL)
        found = chk.check(open(lst).read())
    assert all(v[0] == "k_raw" for v in found), found
    if not found:
        warnings.warn("this toolchain no longer sinks LDS loads behind s_barrier in the minimal reproducer (k_raw is clean)")
    # and no kernel source calls the raw barrier: only workgroup_sync() itself does
    for name in os.listdir(os.path.join(ROOT, "mosfhet_amd", "csrc")):
        if name.endswith((".h", ".inc", ".hip")):
            text = re.sub(r"//.*", "", open(os.path.join(ROOT, "mosfhet_amd", "csrc", name)).read())
            assert text.count("__syncthreads()") == (1 if name == "negacyclic_fft.h" else 0), name


def test_switch_table_matches_the_source():
    """docs/SWITCHES.md (generated by tools/switch_table.py) names every getenv("MOSFHET_...") of the product source and nothing else, and every test it cites exists."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import switch_table
    assert switch_table.in_source() == set(switch_table.SWITCHES), (switch_table.in_source() ^ set(switch_table.SWITCHES))
    assert open(os.path.join(ROOT, "docs", "SWITCHES.md")).read() == switch_table.markdown(), "docs/SWITCHES.md is stale: python tools/switch_table.py --write"
    tests = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    for name, row in switch_table.SWITCHES.items():
        for t in re.findall(r"`(test_\w+)`", row[3]):
            assert "def %s(" % t in tests, (name, t)
    assert len(switch_table.SWITCHES) <= 24


def test_generated_key_switch_block_matches_its_generator(tmp_path):
    """mosfhet_amd/csrc/ks_words_asm.inc (the inline-assembly consume block of the word-lane key switch) is what tools/gen_ks_words_asm.py writes"""
    import subprocess
    import sys
    out = tmp_path / "ks_words_asm.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ks_words_asm.py"), str(out)], check=True)
    assert out.read_text() == open(os.path.join(ROOT, "mosfhet_amd", "csrc", "ks_words_asm.inc")).read(), "ks_words_asm.inc is stale: python tools/gen_ks_words_asm.py"


def test_word_lane_key_switch_plans_fit_the_machine(native_lib):
    """mosfhet_hip_ks_words_plan over a sweep of shapes (no device needed): the stage divides the digit positions, its candidate rows fit three LDS buffers inside the CU's
    160 KiB, a wavefront never has more LDS-DMA requests per stage than the consume block has slots (one per position) nor than vmcnt can count, the grid is a multiple of
    the eight XCDs; digit sets outside 2 - 4 bits, batches below the threshold and seed-compressed LWE rows are left to the other form."""
    import ctypes as C
    lib = native_lib
    lib.mosfhet_hip_ks_words_plan.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_longlong)]
    plan = (C.c_longlong * 8)()
    seen = 0
    for bb in range(1, 9):
        for t in (1, 2, 3, 5, 6, 7, 8, 9, 11, 12, 16, 20, 30):
            if t * bb > 63:
                continue
            for count in (1, 16, 17, 64, 65, 512, 513, 1024, 4096, 9000):
                for n_in, row in ((1, 2), (7, 17), (585, 1025), (1024, 586), (2048, 633), (2048, 4096), (4097, 8192)):
                    assert lib.mosfhet_hip_ks_words_plan(count, n_in, row, t, bb, 0, plan) == 0
                    taken, jb, chunks, pf, lds, wgs, groups, splits = list(plan)
                    assert taken == (1 if (2 <= bb <= 4 and count >= 17) else 0), (bb, t, count, n_in, row)
                    if not 2 <= bb <= 4:
                        continue
                    cands = (1 << bb) - 1
                    assert jb >= 1 and jb * chunks == t and jb * cands <= 96, (bb, t, jb, chunks)
                    assert 1 <= pf <= min(jb, 8) and 16 * pf >= jb * cands, (bb, t, jb, pf)
                    assert lds == 3 * 16 * pf * 512 + 16 + 8192 and lds <= 160 * 1024, (bb, t, lds)
                    assert wgs % 8 == 0 and wgs >= groups * splits and 1 <= splits <= max(1, n_in // 8), (wgs, groups, splits, n_in)
                    assert splits == 1 or wgs <= 256 + 8 * groups, (wgs, groups, splits)      # one round of the chip (the grid is padded to a multiple of 8 per group)
                    assert groups == -(-(-(-min(count, 8192) // 64)) // 8)
                    seen += 1
    assert seen > 1000
    assert lib.mosfhet_hip_ks_words_plan(1024, 2048, 633, 8, 4, 1, plan) == 0 and plan[0] == 0      # seed-compressed LWE rows: the ciphertext-lane tiles
    assert lib.mosfhet_hip_ks_words_plan(0, 1, 1, 1, 2, 0, plan) != 0


def test_no_cpu_fallback(native_lib):
    import torch
    import mosfhet_amd as ma
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ma.MosfhetHipError):
        ma.Engine(0)
    import ctypes as C
    h = C.c_void_p()
    assert native_lib.mosfhet_hip_ctx_create(C.byref(h), 0) != 0
    assert b"no HIP device" in native_lib.mosfhet_hip_last_error() or b"failed" in native_lib.mosfhet_hip_last_error()


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "mosfhet_amd")
    for d, _, files in os.walk(pkg):
        if os.path.basename(d) in ("build", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip")):
                text = open(os.path.join(d, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<].*oracle|liboracle", text, flags=re.M), \
                    os.path.join(d, f)


def test_twiddles_match_oracle(native_lib, oracle):
    from mosfhet_amd import engine
    for N in (512, 1024, 2048, 4096):
        assert (engine.twiddles(N) == oracle.plan(N).twiddles()).all(), N


def test_host_keygen_decrypts_under_the_oracle(native_lib, oracle):
    """Keys and samples from the product's seeded host generator are valid TFHE objects: TRGSW rows decrypt to the
    gadget, TLWE samples to their message, and the LWE key-switch key to s_i * v * 2^(64-(j+1)bb)."""
    from mosfhet_amd import host
    host.seed(42)
    n, N, l, Bg = 6, 1024, 2, 8
    lk, rk = host.LweKey(n, 2.0 ** -30), host.RlweKey(N, 1, 2.0 ** -40)
    assert set(np.unique(lk.s)) <= {0, 1} and set(np.unique(rk.s)) <= {0, 1}
    bk = host.gen_bootstrap_key(rk, lk, l, Bg)
    assert bk.shape == (n, 2 * l, 2, N)
    for i in range(n):
        for p in range(2):
            for j in range(l):
                ph = oracle.trlwe_phase(np.ascontiguousarray(bk[i, p * l + j]), rk.s)
                want = np.zeros(N, dtype=np.uint64)
                # row p*l+j carries s_i * 2^(64-(j+1)Bg) on component p: phase = -s*(that) for p = a, +that for p = b
                h = (int(lk.s[i]) << (64 - (j + 1) * Bg)) % 2 ** 64
                if p == 1:
                    want[0] = h
                    assert oracle.torus_dist(ph, want).max() < 2.0 ** 30
                else:
                    neg_s = (np.uint64(0) - rk.s[0] * np.uint64(h))
                    assert oracle.torus_dist(ph, neg_s).max() < 2.0 ** 30
    msgs = [host.double2torus(m / 8.0) for m in range(8)]
    cts = host.tlwe_samples(msgs, lk)
    assert oracle.torus_dist(host.tlwe_phase(cts, lk.s), np.array(msgs, dtype=np.uint64)).max() < 2.0 ** 40
    assert (host.tlwe_phase(cts, lk.s) == np.array([oracle.tlwe_phase(c, lk.s) for c in cts], dtype=np.uint64)).all()
    out_key = rk.extracted_lwe_key()
    assert (out_key.s == rk.s.reshape(-1)).all()
    ksk = host.gen_tlwe_ks_key(lk, out_key, 3, 2)
    assert ksk.shape == (N, 3, 3, n + 1)
    for i in (0, 5, N - 1):
        for j in range(3):
            for v in (1, 2, 3):
                want = (int(out_key.s[i]) * v << (64 - (j + 1) * 2)) % 2 ** 64
                assert oracle.torus_dist(oracle.tlwe_phase(np.ascontiguousarray(ksk[i, j, v - 1]), lk.s), want) < 2.0 ** 40
    lut = np.array([1, 2, 3, 4], dtype=np.uint64)
    assert (host.torus_packing(lut, 1, N) == oracle.trlwe_torus_packing(lut, 1, N)).all()
    # seeded generator is reproducible
    host.seed(42)
    assert (host.LweKey(n, 2.0 ** -30).s == lk.s).all()


def test_host_circuit_bootstrap_keygen_matches_the_oracle_algorithm(native_lib, oracle):
    """The host layer's private / packing key-switch keys (src/keyswitch.c:39-50,368-390) carry the messages the oracle's
    key switches expect: fed to the ORACLE's trlwe_priv_keyswitch_2 / trlwe_packing1_keyswitch they give the right phases."""
    from mosfhet_amd import host
    host.seed(77)
    N, sigma = 1024, 2.0 ** -44
    rk = host.RlweKey(N, 1, sigma)
    s = rk.s
    ks = host.gen_priv_ks_key(rk, rk, 10, 3)
    assert ks.shape == (2, 10, 2, N)
    r = oracle.Rng(5)
    msg = oracle.u64(r.words(N))
    ct = oracle.trlwe_sample(r, msg, s, sigma)
    out = oracle.trlwe_priv_keyswitch_2(ct, oracle.ks_to_dft(ks[0]), oracle.ks_to_dft(ks[1]), 10, 3)
    want = oracle.poly_naive_mul(np.uint64(0) - s[0], msg)
    assert oracle.torus_dist(oracle.trlwe_phase(out, s), want).max() < 2.0 ** 52
    lk = host.LweKey(24, 2.0 ** -30)
    pk = host.gen_packing1_ks_key(rk, lk, 6, 3)
    assert pk.shape == (24, 6, 7, 2, N)
    c = oracle.tlwe_sample(r, oracle.double2torus(0.125), lk.s, 2.0 ** -30)
    ph = oracle.trlwe_phase(oracle.trlwe_packing1_keyswitch(c, pk, 3), s)
    assert oracle.torus_dist(ph[0], oracle.double2torus(0.125)) < 2.0 ** 50
    assert oracle.torus_dist(ph[1:], np.zeros(N - 1, dtype=np.uint64)).max() < 2.0 ** 50


def test_compat_c_suite_compiles_and_links(native_lib, tmp_path):
    """tests/c/compat_suite.c (a plain C mosfhet.h-style caller, run under -m gpu) builds against the header and library."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "mosfhet_amd")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "compat_suite.c"), "-o", str(tmp_path / "compat_suite"), "-pthread",
                           "-L" + libdir, "-lmosfhet_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])


def test_multi_device_programs_compile_and_link(native_lib, tmp_path):
    """tests/c/multi_device.c (every sharded *_batch entry point against its single calls; run under -m gpu) and tools/in_api_devices.c (BASELINE configs[3] / [4]
    through the drop-in API with a device list) build against the header and library: the sharded entry points, multivalue_bootstrap_CLOT21_batch and
    mosfhet_replication_stats are declared and exported."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "mosfhet_amd")
    for src in (os.path.join(root, "tests", "c", "multi_device.c"), os.path.join(root, "tools", "in_api_devices.c")):
        subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), src, "-o", str(tmp_path / "prog"), "-pthread",
                               "-L" + libdir, "-lmosfhet_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(libdir, "libmosfhet_hip.so")], capture_output=True, text=True, check=True).stdout
    for name in ("mosfhet_hip_bsk_clone", "mosfhet_hip_ksk_clone", "mosfhet_hip_gak_clone", "mosfhet_hip_last_clone_route", "multivalue_bootstrap_CLOT21_batch",
                 "mosfhet_replication_stats"):
        assert (" " + name + "\n") in exported, name


def _build_fileio_helper(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "mosfhet_amd")
    exe = str(tmp_path / "fileio_host")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "c", "fileio_host.c"),
                           "-o", exe, "-L" + libdir, "-lmosfhet_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_on_disk_formats_of_host_objects_are_the_references(native_lib, oracle, tmp_path):
    """tests/golden/fileio.npz holds a file written by the REFERENCE's tlwe_save_key, trlwe_save_key, trgsw_save_key, tlwe_save_sample and
    trlwe_save_sample (tests/golden/make_fileio_golden.py).  The compat readers must take it, and the compat writers must reproduce it byte for byte
    (src/tlwe.c:43-99, src/trlwe.c:24-43,230-251, src/trgsw.c:29-42)."""
    import os
    import struct
    import subprocess
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fileio.npz"))
    n, N, k, l, Bg = (int(g[x]) for x in ("n", "N", "k", "l", "Bg_bit"))
    # the documented layout, spelled out: it must be what the reference wrote
    key_part = struct.pack("<id", n, float(g["lwe_sigma"])) + g["lwe_s"].tobytes()
    rkey = struct.pack("<iid", k, N, float(g["rlwe_sigma"])) + g["rlwe_s"].tobytes()
    expect = key_part + rkey + struct.pack("<ii", l, Bg) + rkey + g["tlwe_ct"].tobytes() + g["trlwe_ct"].tobytes()
    assert g["host_file"].tobytes() == expect
    exe = _build_fileio_helper(tmp_path)
    src, dst = str(tmp_path / "ref.bin"), str(tmp_path / "mine.bin")
    g["host_file"].tofile(src)
    r = subprocess.run([exe, "host", src, dst], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert open(dst, "rb").read() == expect
    fields = dict(kv.split("=") for kv in r.stdout.split())
    assert (int(fields["n"]), int(fields["k"]), int(fields["N"]), int(fields["l"]), int(fields["Bg_bit"])) == (n, k, N, l, Bg)
    assert float(fields["lwe_sigma"]) == float(g["lwe_sigma"]) and float(fields["rlwe_sigma"]) == float(g["rlwe_sigma"])
    assert int(fields["c.b"]) == int(g["tlwe_ct"][n]) and int(fields["rc.b0"]) == int(g["trlwe_ct"][k, 0])
    assert int(fields["phase"]) == int(oracle.tlwe_phase(g["tlwe_ct"], g["lwe_s"]))
    # the LWE key-switch key file (src/tlwe.c:275-287): header n, t, base_bit, n_out, then the table rows a[n_out], b as they lie in the flat layout
    table = g["ks_table"]
    n_in, t, cands, row = table.shape
    assert g["ks_file"].tobytes() == struct.pack("<iiii", n_in, t, int(g["ks_base_bit"]), row - 1) + table.tobytes()
    assert (oracle.tlwe_keyswitch(g["ks_in"], table, row - 1, t, int(g["ks_base_bit"])) == g["ks_switched"]).all()
