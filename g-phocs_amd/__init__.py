"""g-phocs_amd: host-side Python mirror of the C ABI of libgphocs_hip.so (include/gphocs_hip.h),
the MI355X-native per-locus likelihood engine for G-PhoCS-style MCMC.

The package directory name contains a hyphen (it is fixed by the project layout), so import it
through the repo-root alias module `gphocs_amd` (gphocs_amd.py) or importlib.

Python is plumbing only: parsing packs, handing plain buffers to the C ABI, and (multi-GPU)
providing the RCCL all-reduce through torch.distributed.  All computation is in the HIP library;
if the library (or a GPU) is missing, construction fails loudly -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .synth import make_synthetic_pack, replicate_pack, write_pack  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libgphocs_hip.so")
CSRC = os.path.join(_HERE, "csrc")

# -disable-machine-licm: the machine-level loop-invariant code motion hoists constants (a 64-bit 0.0, 1e-6) and
# lane-derived masks out of the proposal loops of the sweep kernel into registers it then has to spill to scratch
# memory -- and reload, hundreds of cycles each, ~10 times per proposal.  Without it the sweep kernel has no vector
# spills at 80 VGPRs (25 before) and 23 instead of 47 scalar spills: -5.4 % sweep time (profiles/HISTORY.md, round 1, v16).
# -structurizecfg-skip-uniform-regions: the backend's CFG structurizer runs on EVERY region by default, also on those
# whose branches are all wave-uniform (nearly all of this code: the chain logic branches on scalars).  Structurizing
# rewrites multi-exit loops and unstructured merges with guard flags (lane masks carried through phis) and pays for
# the extra merges with register copies in the loop bodies; leaving uniform regions as the plain scalar-branch CFG
# they are: -5 % sweep time, -7..18 % static instructions per kernel (profiles/HISTORY.md, round 2).
# -amdgpu-sched-strategy=max-ilp: the sweep kernel is issue-bound at a fixed occupancy (launch bounds), so the
# scheduler has nothing to gain from trading latency hiding for registers: -0.7 % -- for the variants built for 6
# wavefronts per SIMD (80 VGPRs, no spill).  Variant s is built for 8 (64 VGPRs): there the default scheduler spills 18
# vector registers instead of 23 and is the 0.7 % faster one (HIPCC_TUNING_SPILLING).
HIPCC_BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
              "-Wno-unused-result", "-pthread"]
HIPCC_TUNING = ["-mllvm", "-disable-machine-licm", "-mllvm", "-structurizecfg-skip-uniform-regions",
                "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
HIPCC_TUNING_SPILLING = HIPCC_TUNING[:4]
HIPCC_FLAGS = HIPCC_BASE + HIPCC_TUNING
HIPCC_LIBS = ["-ldl", "-lrt"]        # after the sources: an --as-needed linker drops libraries named before their users
# The tuning switches above are backend options of THIS compiler (uniformity analysis decides what
# -structurizecfg-skip-uniform-regions leaves alone; the batched generator's inline asm relies on nothing else in
# the translation unit using M0): the version they were validated with is recorded, build() warns when hipcc
# differs, and a control build WITHOUT them (`libgphocs_hip_plain.so`) runs the golden parity tests next to the
# tuned one (tests/test_gpu_parity.py::test_plain_build_parity), so a compiler change that breaks an assumption
# shows as a difference between the two.
HIPCC_VALIDATED = "roc-7.2.0 26014"
PLAIN_LIB = "libgphocs_hip_plain.so"    # capacities of variant `m`, no -mllvm switches
CHECKED_LIB = "libgphocs_hip_chk.so"    # capacities of variant `x`, -DGPH_BOUNDS: every index of the device code checked (tests only)
CHECKED_LIB_BIG = "libgphocs_hip_chkb.so"   # the same with the capacities of variant `b`: the big-tree (multi-word node sets, list-driven order) and many-band forms

# Capacity variants of the same library (same C ABI, same sources): the static LDS image of a locus is
# sized by compile-time capacities (csrc/gph_types.h), and a tighter image means more resident
# wavefronts per CU.  `load_library(dims=...)` picks the smallest variant that fits the model.
VARIANTS = {  # name: (max leaves, max pops, max bands, waves per SIMD of the sweep kernel, file)
    "s": (16, 9, 4, 8, "libgphocs_hip_s.so"),     # BASELINE configs[0..3]; its image + a 64-pattern block fit 4 LDS granules: 32 loci per CU, 8 wavefronts per SIMD (64 VGPRs, spills outside the inner loops: measured faster than 7 and 6)
    "l": (20, 13, 4, 6, "libgphocs_hip_l.so"),    # BASELINE configs[4] (20 leaves, 13 populations)
    "m": (24, 16, 8, 6, "libgphocs_hip.so"),
    "x": (32, 32, 16, 6, "libgphocs_hip_x.so"),   # the largest variant with one genealogy node per lane (2n-1 <= 63) and 32-bit population sets
    # the engine's hard caps: 64 leaves, the reference's own 39 populations (NSPECIES 20), 16 bands -- 128-bit node sets,
    # 64-bit population sets, 16-bit event ids, list-driven pruning order instead of the lane-per-node wave programs
    "g": (48, 16, 8, 6, "libgphocs_hip_g.so"),    # many samples, few populations (e.g. 20 diploids over 5 populations): the big-tree forms with a 9-KB image
    "h": (64, 40, 16, 6, "libgphocs_hip_h.so"),
    # more than 16 migration bands, up to the reference's MAX_MIG_BANDS 100 (patch.h:17): the live-band list of a chain walk in LDS
    # instead of 16 nibbles of a scalar, the model read from the chain state in HBM by every kernel (it no longer fits the
    # kernel-argument segment), 384-column reduced rows; with the 64-leaf / 39-population forms of `h`: the engine's hard caps
    "b": (64, 40, 100, 6, "libgphocs_hip_b.so"),
    # the reference's own compile-time caps (NS 200, 39 populations, MAX_MIG_BANDS 100: patch.h:17-22): node sets of seven 64-bit
    # words, the scalar pad in two registers, a 44-KB LDS image (3 loci resident per CU) -- a capability build
    "n": (200, 40, 100, 2, "libgphocs_hip_n.so"),
}


LIB_SOURCES = ("gph_engine.hip", "gph_mcmc.cpp", "gph_input.cpp", "gph_program.cpp", "gph_readtrace.cpp", "gph_comm.cpp")


def variant_for(n, K, B):
    for name in ("s", "l", "m", "x", "g", "h", "b", "n"):
        cl, ck, cb, _, _ = VARIANTS[name]
        if n <= cl and K <= ck and B <= cb:
            return name
    raise RuntimeError(f"model (leaves={n}, pops={K}, bands={B}) exceeds the engine's caps (200 leaves, 39 populations, "
                       f"100 bands); the reference's own compile-time caps are "
                       f"200 leaves / 39 populations / 100 bands (upstream src/patch.h:17-22)")


def lib_path(name="m"):
    return os.path.join(_HERE, VARIANTS[name][4])


def build(verbose=False):
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU), every variant.  Safe to call from
    several processes at once (one rank per GPU all call it): an exclusive file lock serialises them, the first one
    builds what is out of date into a temporary file and renames it, the others find everything up to date."""
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _run_to(cmd, out, verbose):
    tmp = f"{out}.tmp.{os.getpid()}"
    if verbose:
        print(" ".join(cmd + ["-o", out]))
    try:
        subprocess.run(cmd + ["-o", tmp], check=True)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


def hipcc_version():
    try:
        out = subprocess.run(["hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:  # pragma: no cover
        return ""
    for ln in out.splitlines():
        if "clang version" in ln:
            return ln.strip()
    return out.strip().splitlines()[0] if out.strip() else ""


def _build_locked(verbose):
    ver = hipcc_version()
    if ver and HIPCC_VALIDATED not in ver:
        print(f"gphocs_amd.build: hipcc is '{ver}', the backend switches {HIPCC_TUNING[1::2]} were validated with "
              f"'{HIPCC_VALIDATED}': run the -m gpu parity suite (it compares the tuned build with the plain one)")
    srcs = [os.path.join(CSRC, f) for f in LIB_SOURCES]
    import hashlib
    hashed = sorted(set(LIB_SOURCES) | {f for f in os.listdir(CSRC) if f.endswith(".h")})

    def build_id(flags):
        """sources + flags: what the binary is a function of besides the compiler.  The compiler's version is recorded next
        to it (sidecar, bench line) and NOT hashed: a library that travelled to a box with another hipcc -- or none -- is
        still the build of these sources, and recompiling it there would put an unvalidated compiler's code (and minutes
        of build time) inside a measurement.  Only the translation units and headers count: an editor's backup file in
        csrc/ does not make the libraries stale."""
        h = hashlib.sha256()
        for f in hashed:
            h.update(f.encode() + b"\0" + open(os.path.join(CSRC, f), "rb").read())
        h.update(" ".join(flags).encode())
        return h.hexdigest()[:12]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(REPO, "include", "gphocs_hip.h")]
    hdr = hashlib.sha256(open(deps[-1], "rb").read()).hexdigest()[:8]

    def library(out, tag, fl):
        """up to date = built from exactly these sources with these flags: the id the library was built with sits in a
        sidecar file next to it (time stamps do not survive checkouts and snapshot copies reliably, and a rebuild on the
        GPU box costs minutes of the measurement budget); the sidecar's third field is the compiler that built it"""
        want = f"{tag}-{build_id(fl)}"
        side = out + ".buildid"
        try:
            if os.path.exists(out):
                got = open(side).read().strip().split(":", 2)
                if got[:2] == [want, hdr]:
                    built_by = got[2] if len(got) > 2 else ""
                    if ver and built_by and built_by != ver and verbose:
                        print(f"gphocs_amd.build: {os.path.basename(out)} was built by '{built_by}', this box has '{ver}': kept")
                    return
        except OSError:
            pass
        if not ver:
            if os.path.exists(out):
                print(f"gphocs_amd.build: no hipcc here; using the existing {os.path.basename(out)} although its build id "
                      f"cannot be confirmed as {want}")
                return
            raise RuntimeError(f"gphocs_amd.build: hipcc not found and {out} does not exist")
        _run_to(["hipcc"] + fl + [f'-DGPH_BUILD_ID="{want}"'] + srcs + HIPCC_LIBS, out, verbose)
        with open(side, "w") as f:
            f.write(f"{want}:{hdr}:{ver}\n")

    jobs = []
    for name, (cl, ck, cb, waves, fn) in VARIANTS.items():
        jobs.append((os.path.join(_HERE, fn), name,
                     HIPCC_BASE + (HIPCC_TUNING_SPILLING if waves > 6 else HIPCC_TUNING) +
                     [f"-DGPH_CAP_LEAVES={cl}", f"-DGPH_CAP_K={ck}", f"-DGPH_CAP_B={cb}", f"-DGPH_SWEEP_WAVES={waves}"]))
    # control build without the backend switches (parity tests only)
    cl, ck, cb, waves, _ = VARIANTS["m"]
    jobs.append((os.path.join(_HERE, PLAIN_LIB), "plain",
                 HIPCC_BASE + [f"-DGPH_CAP_LEAVES={cl}", f"-DGPH_CAP_K={ck}", f"-DGPH_CAP_B={cb}", f"-DGPH_SWEEP_WAVES={waves}",
                               "-DGPH_LOGSTEPS"]))     # + the decision-level transcript (tests/test_logsteps.py)
    # checked build (tests only): every index of the per-locus device code against its array's extent (gph_rt.h: GPH_BOUNDS)
    cl, ck, cb, waves, _ = VARIANTS["x"]
    jobs.append((os.path.join(_HERE, CHECKED_LIB), "chk",
                 HIPCC_BASE + HIPCC_TUNING + [f"-DGPH_CAP_LEAVES={cl}", f"-DGPH_CAP_K={ck}", f"-DGPH_CAP_B={cb}", f"-DGPH_SWEEP_WAVES={waves}",
                                              "-DGPH_BOUNDS"]))
    cl, ck, cb, waves, _ = VARIANTS["b"]
    jobs.append((os.path.join(_HERE, CHECKED_LIB_BIG), "chkb",
                 HIPCC_BASE + HIPCC_TUNING + [f"-DGPH_CAP_LEAVES={cl}", f"-DGPH_CAP_K={ck}", f"-DGPH_CAP_B={cb}", f"-DGPH_SWEEP_WAVES={waves}",
                                              "-DGPH_BOUNDS"]))
    # every variant is one hipcc process of its own (half a minute each): a few at a time
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(5, (os.cpu_count() or 2) // 2))) as ex:
        for f in [ex.submit(library, *j) for j in jobs]:
            f.result()
    # the program: same command line as the reference's G-PhoCS binary (GPhoCS.c:84-238)
    exe, main = os.path.join(_HERE, "G-PhoCS-hip"), os.path.join(CSRC, "gph_main.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(main), os.path.getmtime(deps[-1])):
        _run_to(["g++", "-O2", "-std=c++17", "-I", os.path.join(REPO, "include"), main, "-ldl"], exe, verbose)
    # the post-run trace summary tool (readTrace.c), host only
    exe, main = os.path.join(_HERE, "readTrace"), os.path.join(CSRC, "gph_readtrace.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(main), os.path.getmtime(deps[-1])):
        _run_to(["g++", "-O2", "-std=c++17", "-DGPH_READTRACE_MAIN", main], exe, verbose)
    return LIB_PATH


# ------------------------------------------------------------------------------------ C types
class GphConfig(C.Structure):
    _fields_ = [("n", C.c_int32), ("Kc", C.c_int32), ("K", C.c_int32), ("B", C.c_int32),
                ("rootPop", C.c_int32),
                ("samplesPerPop", C.POINTER(C.c_int32)), ("popFather", C.POINTER(C.c_int32)),
                ("popSon0", C.POINTER(C.c_int32)), ("popSon1", C.POINTER(C.c_int32)),
                ("bandSrc", C.POINTER(C.c_int32)), ("bandTgt", C.POINTER(C.c_int32)),
                ("device", C.c_int32), ("L_total", C.c_int64), ("locus_begin", C.c_int64)]


class GphSweepResult(C.Structure):
    _fields_ = [("accepted_internal", C.c_int64), ("accepted_mignode", C.c_int64),
                ("accepted_spr", C.c_int64), ("dData_internal", C.c_double),
                ("dLog_internal", C.c_double), ("dLog_mignode", C.c_double),
                ("dData_spr", C.c_double), ("dLog_spr", C.c_double), ("total_mig_nodes", C.c_int64)]


class GphCounters(C.Structure):
    _fields_ = [("evals", C.c_int64), ("eval_nodes", C.c_int64), ("eval_bytes", C.c_double),
                ("not_enough_migs", C.c_int64)]


class GphMcmcConfig(C.Structure):
    _fields_ = [("thetaAlpha", C.POINTER(C.c_double)), ("thetaBeta", C.POINTER(C.c_double)),
                ("thetaStart", C.POINTER(C.c_double)), ("ageAlpha", C.POINTER(C.c_double)),
                ("ageBeta", C.POINTER(C.c_double)), ("ageStart", C.POINTER(C.c_double)),
                ("sampleAge", C.POINTER(C.c_double)), ("updateSampleAge", C.POINTER(C.c_int32)),
                ("mrAlpha", C.POINTER(C.c_double)),
                ("mrBeta", C.POINTER(C.c_double)),
                ("ftCoalTime", C.c_double), ("ftMigTime", C.c_double), ("ftTheta", C.c_double),
                ("ftMigRate", C.c_double), ("ftMixing", C.c_double),
                ("ftTaus", C.POINTER(C.c_double)),
                ("seed", C.c_int32), ("startMig", C.c_int32), ("doMixing", C.c_int32),
                ("samplesPerLog", C.c_int32), ("numParameters", C.c_int32),
                ("printFactors", C.POINTER(C.c_double)),
                ("mutRateMode", C.c_int32), ("varRatesAlpha", C.c_double), ("ftLocusRate", C.c_double)]


class GphControlInfo(C.Structure):
    _fields_ = [("seqFile", C.c_char_p), ("traceFile", C.c_char_p), ("rateFile", C.c_char_p),
                ("numLoci", C.c_int32), ("burnin", C.c_int32), ("numSamples", C.c_int32),
                ("sampleSkip", C.c_int32), ("logsPerLine", C.c_int32), ("mutRateMode", C.c_int32),
                ("findFinetunes", C.c_int32), ("findFinetunesNumSteps", C.c_int32),
                ("findFinetunesSamplesPerStep", C.c_int32), ("numSampleSlots", C.c_int32),
                ("varRatesAlpha", C.c_double), ("ftLocusRate", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32,
                           C.POINTER(C.c_double), C.c_int32)

EXPORTS = [  # every symbol include/gphocs_hip.h declares
    "gph_engine_create", "gph_engine_destroy", "gph_engine_set_allreduce", "gph_engine_load_loci",
    "gph_engine_set_model", "gph_engine_seed", "gph_engine_init_genealogies",
    "gph_engine_genealogy_sweep", "gph_engine_tau_evaluate", "gph_engine_tau_commit",
    "gph_engine_tau_revert", "gph_engine_mixing_evaluate", "gph_engine_mixing_commit",
    "gph_engine_mixing_revert", "gph_engine_apply_theta", "gph_engine_apply_migrate",
    "gph_engine_get_totals", "gph_engine_synchronize", "gph_engine_check_all",
    "gph_engine_get_counters", "gph_engine_dump_loci", "gph_engine_last_kernel_ms",
    "gph_engine_num_loci", "gph_engine_hbm_bytes", "gph_debug_math", "gph_engine_class_stats",
    "gph_mcmc_create", "gph_mcmc_destroy", "gph_mcmc_initialize", "gph_mcmc_set_record_file",
    "gph_mcmc_iteration", "gph_mcmc_get_state", "gph_mcmc_dump_state", "gph_mcmc_accept_counts",
    "gph_mcmc_param_vals", "gph_mcmc_tau_accept_counts", "gph_mcmc_set_finetunes", "gph_mcmc_set_log_period", "gph_engine_last_error", "gph_engine_debug_break_chain", "gph_engine_debug_oob", "gph_comm_peer_exchange", "gph_comm_peer_next_gen", "gph_control_read", "gph_control_free", "gph_control_get", "gph_control_pop_name",
    "gph_control_sample_name", "gph_loci_read", "gph_loci_free", "gph_loci_arrays", "gph_run_control_file", "gph_run_control_file_ranked",
    "gph_read_trace", "gph_engine_locus_rate_update", "gph_engine_set_locus_rates", "gph_mcmc_set_locus_rate_finetune",
    "gph_mcmc_locus_rate_state", "gph_engine_set_comm", "gph_engine_host_stats", "gph_engine_set_timing",
    "gph_comm_unique_id", "gph_comm_create_rccl", "gph_comm_create_shm", "gph_comm_attach_shm", "gph_comm_shm_bytes",
    "gph_comm_destroy", "gph_comm_world", "gph_comm_rank", "gph_comm_on_stream", "gph_comm_kind",
    "gph_comm_allgather_stream", "gph_comm_allreduce_host", "gph_run_control_file_comm", "gph_device_count",
    "gph_engine_unit", "gph_build_id", "gph_build_compiler", "gph_runtime_version", "gph_engine_steplog_enable", "gph_engine_steplog_fetch", "gph_comm_local_group", "gph_comm_create_local",
    "gph_mcmc_get_chain", "gph_mcmc_set_chain", "gph_mcmc_update_gb", "gph_mcmc_update_locus_rate", "gph_mcmc_update_theta",
    "gph_mcmc_update_mig_rates", "gph_mcmc_update_tau", "gph_mcmc_update_sample_age", "gph_mcmc_mixing",
    "gph_mcmc_synchronize_events", "gph_mcmc_check_all", "gph_mcmc_initialize_genealogies",
]


_LIBS = {}


def load_library(path=None, dims=None):
    """dims = (leaves, pops, bands): load the tightest capacity variant that fits"""
    if path is None and dims is not None:
        path = lib_path(variant_for(*dims))
    path = path or LIB_PATH
    if path in _LIBS:
        return _LIBS[path]
    lib = _load_library(path)
    _LIBS[path] = lib
    return lib


def _load_library(path):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if an ABI symbol is missing
    lib.gph_engine_num_loci.restype = C.c_int64
    lib.gph_engine_destroy.restype = None
    lib.gph_mcmc_destroy.restype = None
    lib.gph_engine_seed.argtypes = [C.c_void_p, C.c_uint32]
    lib.gph_engine_tau_revert.argtypes = [C.c_void_p, C.c_int64]
    lib.gph_engine_mixing_evaluate.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double)]
    lib.gph_engine_mixing_commit.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.gph_engine_genealogy_sweep.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                               C.POINTER(GphSweepResult)]
    lib.gph_engine_apply_theta.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double]
    lib.gph_engine_apply_migrate.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double]
    lib.gph_engine_load_loci.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    lib.gph_mcmc_iteration.argtypes = [C.c_void_p, C.c_int32]
    lib.gph_mcmc_dump_state.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    lib.gph_engine_dump_loci.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]
    lib.gph_engine_last_error.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.gph_engine_debug_break_chain.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    lib.gph_mcmc_set_record_file.argtypes = [C.c_void_p, C.c_char_p]
    lib.gph_engine_last_kernel_ms.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
    lib.gph_engine_get_counters.argtypes = [C.c_void_p, C.POINTER(GphCounters), C.c_int32]
    lib.gph_engine_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
    lib.gph_mcmc_param_vals.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.gph_mcmc_tau_accept_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.gph_mcmc_set_finetunes.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.c_void_p]
    lib.gph_control_read.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    lib.gph_control_free.argtypes = [C.c_void_p]
    lib.gph_control_free.restype = None
    lib.gph_control_get.argtypes = [C.c_void_p, C.POINTER(GphConfig), C.POINTER(GphMcmcConfig),
                                    C.POINTER(GphControlInfo)]
    lib.gph_control_pop_name.argtypes = [C.c_void_p, C.c_int32]
    lib.gph_control_pop_name.restype = C.c_char_p
    lib.gph_control_sample_name.argtypes = [C.c_void_p, C.c_int32]
    lib.gph_control_sample_name.restype = C.c_char_p
    lib.gph_loci_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_void_p), C.c_char_p, C.c_int32]
    lib.gph_loci_free.argtypes = [C.c_void_p]
    lib.gph_loci_free.restype = None
    lib.gph_loci_arrays.argtypes = [C.c_void_p] + [C.c_void_p] * 8
    lib.gph_run_control_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32]
    lib.gph_run_control_file_ranked.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                ALLREDUCE_FN, C.c_void_p]
    lib.gph_comm_unique_id.argtypes = [C.c_void_p]
    lib.gph_comm_create_rccl.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.gph_comm_create_rccl.restype = C.c_void_p
    lib.gph_comm_create_shm.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
    lib.gph_comm_create_shm.restype = C.c_void_p
    lib.gph_comm_destroy.argtypes = [C.c_void_p]
    lib.gph_comm_destroy.restype = None
    lib.gph_comm_kind.argtypes = [C.c_void_p]
    lib.gph_comm_kind.restype = C.c_char_p
    lib.gph_engine_set_comm.argtypes = [C.c_void_p, C.c_void_p]
    lib.gph_engine_host_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int32)]
    lib.gph_engine_set_timing.argtypes = [C.c_void_p, C.c_uint32]
    lib.gph_engine_unit.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_int32]
    lib.gph_run_control_file_comm.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.gph_build_id.restype = C.c_char_p
    lib.gph_build_compiler.restype = C.c_char_p
    lib.gph_engine_steplog_enable.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32]
    lib.gph_engine_steplog_fetch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    lib.gph_runtime_version.restype = C.c_char_p
    lib.gph_comm_local_group.argtypes = [C.c_int32, C.c_int32]
    lib.gph_comm_local_group.restype = C.c_void_p
    lib.gph_comm_create_local.argtypes = [C.c_void_p, C.c_int32]
    lib.gph_comm_create_local.restype = C.c_void_p
    lib.gph_comm_world.argtypes = [C.c_void_p]
    lib.gph_comm_rank.argtypes = [C.c_void_p]
    lib.gph_comm_on_stream.argtypes = [C.c_void_p]
    lib.gph_comm_attach_shm.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.gph_comm_attach_shm.restype = C.c_void_p
    return lib


# ------------------------------------------------------------------------------------ packs
def encode_phases(counts):
    """phase counts as the C ABI carries them (include/gphocs_hip.h: GPH_NUMPHASES): below 2^15 as they are, a power of two from
    2^15 on as 0x8000 | exponent"""
    a = np.asarray(counts, dtype=np.int64)
    big = a >= 0x8000
    if big.any():
        ex = np.round(np.log2(np.where(big, a, 1))).astype(np.int64)
        assert np.all((1 << ex[big]) == a[big]), "a phase count of 2^15 or more must be a power of two"
        a = np.where(big, 0x8000 | ex, a)
    return a.astype(np.uint16)


def decode_phases(words):
    w = np.asarray(words, dtype=np.int64)
    return np.where(w < 0x8000, w, np.left_shift(1, w & 31))


class Pack:
    """Model + processed loci (the input of gph_engine_load_loci): what the reference holds
    after readControlFile + processAlignments (GPhoCS.c:147-235).  Text format written by
    oracle/ref_harness.c `pack`; synthetic packs for the benchmark come from make_synthetic()."""

    CODES = {"T": 0, "C": 1, "A": 2, "G": 3, "N": 4}

    def __init__(self):
        self.L = 0

    @staticmethod
    def load(path):
        p = Pack()
        with open(path) as f:
            tok = f.read().split()
        it = iter(tok)

        def nxt():
            return next(it)

        def fl():
            return float.fromhex(nxt())
        assert nxt() == "GPHOCS-PACK" and nxt() == "1"
        for key in ("numLoci", "numSamples", "numCurPops", "numPops", "numMigBands", "rootPop"):
            assert nxt() == key
            setattr(p, key, int(nxt()))
        p.L, p.n, p.Kc, p.K, p.B = p.numLoci, p.numSamples, p.numCurPops, p.numPops, p.numMigBands
        assert nxt() == "samplesPerPop"
        p.samplesPerPop = np.array([int(nxt()) for _ in range(p.Kc)], dtype=np.int32)
        K, B = p.K, p.B
        p.popName = []
        p.popFather = np.zeros(K, np.int32)
        p.popSon0 = np.zeros(K, np.int32)
        p.popSon1 = np.zeros(K, np.int32)
        p.sampleAge = np.zeros(K)
        p.updateSampleAge = np.zeros(K, np.int32)
        p.thetaAlpha, p.thetaBeta, p.thetaStart = np.zeros(K), np.zeros(K), np.zeros(K)
        p.ageAlpha, p.ageBeta, p.ageStart = np.zeros(K), np.zeros(K), np.zeros(K)
        for k in range(K):
            assert nxt() == "pop" and int(nxt()) == k
            p.popName.append(nxt())
            p.popFather[k], p.popSon0[k], p.popSon1[k] = int(nxt()), int(nxt()), int(nxt())
            p.sampleAge[k] = fl()
            p.updateSampleAge[k] = int(nxt())
            p.thetaAlpha[k], p.thetaBeta[k], p.thetaStart[k] = fl(), fl(), fl()
            p.ageAlpha[k], p.ageBeta[k], p.ageStart[k] = fl(), fl(), fl()
        p.bandSrc, p.bandTgt = np.zeros(max(B, 1), np.int32), np.zeros(max(B, 1), np.int32)
        p.mrAlpha, p.mrBeta = np.zeros(max(B, 1)), np.ones(max(B, 1))
        for b in range(B):
            assert nxt() == "band" and int(nxt()) == b
            p.bandSrc[b], p.bandTgt[b] = int(nxt()), int(nxt())
            p.mrAlpha[b], p.mrBeta[b] = fl(), fl()
        assert nxt() == "mcmc"
        (p.seed, p.burnin, p.numSamplesMcmc, p.sampleSkip, p.startMig, p.doMixing, p.samplesPerLog,
         p.mutRateMode) = [int(nxt()) for _ in range(8)]
        assert nxt() == "finetunes"
        p.ftCoalTime, p.ftMigTime, p.ftTheta, p.ftMigRate, p.ftMixing = fl(), fl(), fl(), fl(), fl()
        p.ftTaus = np.array([fl() for _ in range(K)])
        key = nxt()
        p.varRatesAlpha, p.ftLocusRate = 1.0, -1.0
        if key == "locusrate":            # packs of `locus-mut-rate VAR` control files only
            p.varRatesAlpha, p.ftLocusRate = fl(), fl()
            key = nxt()
        assert key == "printFactors"
        p.numParameters = int(nxt())
        p.printFactors = np.array([fl() for _ in range(p.numParameters)])
        offs, leaf, phases, counts, rates = [0], [], [], [], []
        for g in range(p.L):
            assert nxt() == "locus" and int(nxt()) == g
            P = int(nxt())
            rates.append(fl())
            for _ in range(P):
                s = nxt()
                assert len(s) == p.n
                leaf.append([Pack.CODES[ch] for ch in s])
                phases.append(int(nxt()))
                counts.append(int(nxt()))
            offs.append(offs[-1] + P)
        p.pattern_offsets = np.array(offs, dtype=np.int64)
        p.leafcodes = np.array(leaf, dtype=np.uint8).reshape(-1, p.n)
        p.numPhases = encode_phases(phases)
        p.counts = np.array(counts, dtype=np.int32)
        p.mutRates = np.array(rates)
        return p

    @staticmethod
    def from_control(ctl_path, lib=None, secondary=None, seq_path=None, threads=0):
        """control file + sequence file -> Pack, through the library's own front end
        (gph_control_read / gph_loci_read; the reference's readControlFile / readSeqFile path)"""
        lib = lib or load_library()
        ctl = C.c_void_p()
        rc = lib.gph_control_read(os.fsencode(ctl_path), os.fsencode(secondary) if secondary else None, C.byref(ctl))
        if rc:
            raise ValueError(f"gph_control_read({ctl_path}) failed with status {rc}")
        try:
            cfg, mc, info = GphConfig(), GphMcmcConfig(), GphControlInfo()
            lib.gph_control_get(ctl, C.byref(cfg), C.byref(mc), C.byref(info))
            p = Pack()
            p.n, p.Kc, p.K, p.B, p.rootPop = cfg.n, cfg.Kc, cfg.K, cfg.B, cfg.rootPop
            K, B = p.K, p.B

            def arr(ptr, k, dt):
                return np.array([ptr[i] for i in range(k)], dtype=dt)
            p.samplesPerPop = arr(cfg.samplesPerPop, p.Kc, np.int32)
            p.popFather, p.popSon0, p.popSon1 = (arr(cfg.popFather, K, np.int32), arr(cfg.popSon0, K, np.int32),
                                                 arr(cfg.popSon1, K, np.int32))
            p.bandSrc = arr(cfg.bandSrc, B, np.int32) if B else np.zeros(1, np.int32)
            p.bandTgt = arr(cfg.bandTgt, B, np.int32) if B else np.zeros(1, np.int32)
            p.popName = [lib.gph_control_pop_name(ctl, k).decode() for k in range(K)]
            p.sampleNames = [lib.gph_control_sample_name(ctl, i).decode() for i in range(p.n)]
            for nm in ("thetaAlpha", "thetaBeta", "thetaStart", "ageAlpha", "ageBeta", "ageStart", "sampleAge", "ftTaus"):
                setattr(p, nm, arr(getattr(mc, nm), K, np.float64))
            p.updateSampleAge = arr(mc.updateSampleAge, K, np.int32)
            p.mrAlpha = arr(mc.mrAlpha, B, np.float64) if B else np.zeros(1)
            p.mrBeta = arr(mc.mrBeta, B, np.float64) if B else np.ones(1)
            p.ftCoalTime, p.ftMigTime, p.ftTheta, p.ftMigRate, p.ftMixing = (mc.ftCoalTime, mc.ftMigTime, mc.ftTheta,
                                                                             mc.ftMigRate, mc.ftMixing)
            p.seed, p.startMig, p.doMixing, p.samplesPerLog = mc.seed, mc.startMig, mc.doMixing, mc.samplesPerLog
            p.burnin, p.numSamplesMcmc, p.sampleSkip, p.mutRateMode = (info.burnin, info.numSamples, info.sampleSkip,
                                                                       info.mutRateMode)
            p.varRatesAlpha, p.ftLocusRate = info.varRatesAlpha, info.ftLocusRate
            p.numParameters = mc.numParameters
            p.printFactors = arr(mc.printFactors, p.numParameters, np.float64)
            p.traceFile, p.seqFile = info.traceFile.decode(), info.seqFile.decode()
            loci = C.c_void_p()
            err = C.create_string_buffer(512)
            rc = lib.gph_loci_read(ctl, os.fsencode(seq_path) if seq_path else None, threads, C.byref(loci), err, 512)
            if rc:
                raise ValueError(f"gph_loci_read failed with status {rc}: {err.value.decode()}")
            try:
                L, n = C.c_int64(), C.c_int32()
                po, lf, ph, cn, mr, up = (C.c_void_p() for _ in range(6))
                lib.gph_loci_arrays(loci, C.byref(L), C.byref(n), C.byref(po), C.byref(lf), C.byref(ph), C.byref(cn),
                                    C.byref(mr), C.byref(up))
                p.L = p.numLoci = L.value

                def view(ptr, count, ct, dt):
                    if count == 0:
                        return np.zeros(0, dt)
                    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(count,)).astype(dt, copy=True)
                p.pattern_offsets = view(po, p.L + 1, C.c_int64, np.int64)
                Ptot = int(p.pattern_offsets[-1])
                p.leafcodes = view(lf, Ptot * p.n, C.c_uint8, np.uint8).reshape(-1, p.n)
                p.numPhases = view(ph, Ptot, C.c_uint16, np.uint16)
                p.counts = view(cn, Ptot, C.c_int32, np.int32)
                p.mutRates = view(mr, p.L, C.c_double, np.float64)
                p.unphased = view(up, p.L, C.c_int32, np.int32)
            finally:
                lib.gph_loci_free(loci)
            return p
        finally:
            lib.gph_control_free(ctl)

    def shard(self, rank, world):
        """contiguous block of ceil(L/world) loci (mirrors OpenMP static scheduling,
        MultiCoreUtils.h:8); returns (begin, end)"""
        per = (self.L + world - 1) // world
        return min(rank * per, self.L), min((rank + 1) * per, self.L)


def _dp(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))


class Sampler:
    """Engine + host MCMC driver for one rank's shard of a Pack."""

    def __init__(self, pack, lib=None, device=0, rank=0, world=1, allreduce=None, comm=None):
        """allreduce: Python hook (gloo tests) -- forces a host synchronisation per reduction point;
        comm: native communicator handle (gph_comm_create_rccl / _shm) -- RCCL runs on the engine's stream"""
        self.lib = lib or load_library(dims=(pack.n, pack.K, pack.B))
        self.pack = pack
        self.rank, self.world = rank, world
        self.begin, self.end = pack.shard(rank, world)
        p = pack
        # a pack that already holds only this rank's loci (bench.py, weak scaling): global_L loci over
        # all ranks, this rank's first locus at global_begin
        L_total, g_begin = p.L, self.begin
        if getattr(p, "global_L", None) is not None:
            L_total, g_begin = int(p.global_L), int(p.global_begin)
            self.begin, self.end = 0, p.L
        self._keep = dict(spp=np.ascontiguousarray(p.samplesPerPop, np.int32),
                          pf=np.ascontiguousarray(p.popFather, np.int32),
                          s0=np.ascontiguousarray(p.popSon0, np.int32),
                          s1=np.ascontiguousarray(p.popSon1, np.int32),
                          bs=np.ascontiguousarray(p.bandSrc, np.int32),
                          bt=np.ascontiguousarray(p.bandTgt, np.int32))
        k = self._keep
        self.cfg = GphConfig(p.n, p.Kc, p.K, p.B, p.rootPop, _ip(k["spp"]), _ip(k["pf"]), _ip(k["s0"]),
                             _ip(k["s1"]), _ip(k["bs"]), _ip(k["bt"]), device, L_total, g_begin)
        self.engine = C.c_void_p()
        self._chk(self.lib.gph_engine_create(C.byref(self.cfg), C.byref(self.engine)), "engine_create")
        self._cb = None
        if allreduce is not None:
            def _cb(user, sums, nsum, mins, nmin):
                try:
                    s = np.ctypeslib.as_array(sums, shape=(nsum,)) if nsum else np.zeros(0)
                    m = np.ctypeslib.as_array(mins, shape=(nmin,)) if nmin else np.zeros(0)
                    allreduce(s, m)
                    return 0
                except Exception as ex:  # pragma: no cover
                    print("allreduce callback failed:", ex)
                    return 1
            self._cb = ALLREDUCE_FN(_cb)
            self._chk(self.lib.gph_engine_set_allreduce(self.engine, self._cb, None), "set_allreduce")
        if comm is not None:
            self._chk(self.lib.gph_engine_set_comm(self.engine, comm), "set_comm")
        b, e = self.begin, self.end
        o0, o1 = p.pattern_offsets[b], p.pattern_offsets[e]
        offs = np.ascontiguousarray(p.pattern_offsets[b:e + 1] - o0, dtype=np.int64)
        leaf = np.ascontiguousarray(p.leafcodes[o0:o1])
        ph = np.ascontiguousarray(p.numPhases[o0:o1], dtype=np.uint16)
        cn = np.ascontiguousarray(p.counts[o0:o1])
        rates = np.ascontiguousarray(p.mutRates[b:e], dtype=np.float64)
        use_rates = bool(np.any(rates != 1.0))
        self._chk(self.lib.gph_engine_load_loci(self.engine, e - b, offs.ctypes.data, leaf.ctypes.data,
                                                ph.ctypes.data, cn.ctypes.data,
                                                rates.ctypes.data if use_rates else None), "load_loci")
        k.update(ta=p.thetaAlpha, tb=p.thetaBeta, ts=p.thetaStart, aa=p.ageAlpha, ab=p.ageBeta,
                 as_=p.ageStart, sa=p.sampleAge, ma=p.mrAlpha, mb=p.mrBeta, ft=p.ftTaus, pf_=p.printFactors,
                 usa=np.ascontiguousarray(getattr(p, "updateSampleAge", np.zeros(p.K, np.int32)), dtype=np.int32))
        self.mcfg = GphMcmcConfig(_dp(k["ta"]), _dp(k["tb"]), _dp(k["ts"]), _dp(k["aa"]), _dp(k["ab"]),
                                  _dp(k["as_"]), _dp(k["sa"]), k["usa"].ctypes.data_as(C.POINTER(C.c_int32)),
                                  _dp(k["ma"]), _dp(k["mb"]),
                                  p.ftCoalTime, p.ftMigTime, p.ftTheta, p.ftMigRate, p.ftMixing,
                                  _dp(k["ft"]), p.seed, p.startMig, p.doMixing, p.samplesPerLog,
                                  p.numParameters, _dp(k["pf_"]),
                                  int(getattr(p, "mutRateMode", 0)) if int(getattr(p, "mutRateMode", 0)) == 1 else 0,
                                  float(getattr(p, "varRatesAlpha", 1.0)), float(getattr(p, "ftLocusRate", -1.0)))
        self.mcmc = C.c_void_p()
        self._chk(self.lib.gph_mcmc_create(self.engine, C.byref(self.cfg), C.byref(self.mcfg),
                                           C.byref(self.mcmc)), "mcmc_create")

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"gphocs_hip: {what} failed with status {rc}")

    def set_record_file(self, path):
        self._chk(self.lib.gph_mcmc_set_record_file(self.mcmc, path.encode() if path else None), "record")

    def initialize(self):
        tc = C.c_int64()
        self._chk(self.lib.gph_mcmc_initialize(self.mcmc, C.byref(tc)), "initialize")
        return tc.value

    def iteration(self, it):
        self._chk(self.lib.gph_mcmc_iteration(self.mcmc, it), f"iteration {it}")

    def state(self):
        ll, dl = C.c_double(), C.c_double()
        th, ag, mr = np.zeros(self.pack.K), np.zeros(self.pack.K), np.zeros(max(self.pack.B, 1))
        self._chk(self.lib.gph_mcmc_get_state(self.mcmc, C.byref(ll), C.byref(dl), _dp(th), _dp(ag), _dp(mr)),
                  "get_state")
        return dict(logLikelihood=ll.value, dataLogLikelihood=dl.value, theta=th, popAge=ag,
                    migRate=mr[:self.pack.B])

    def dump_state(self, path, with_cond=False):
        self._chk(self.lib.gph_mcmc_dump_state(self.mcmc, path.encode(), int(with_cond)), "dump_state")

    def last_error(self):
        """(global locus index or -1, reference "Fatal Error" code) of the last call that failed with GPH_EKERNEL"""
        g, c = C.c_int64(-1), C.c_int32(0)
        self._chk(self.lib.gph_engine_last_error(self.engine, C.byref(g), C.byref(c)), "last_error")
        return g.value, c.value

    def counters(self, reset=False):
        c = GphCounters()
        self._chk(self.lib.gph_engine_get_counters(self.engine, C.byref(c), int(reset)), "counters")
        return dict(evals=c.evals, eval_nodes=c.eval_nodes, eval_bytes=c.eval_bytes,
                    not_enough_migs=c.not_enough_migs)

    def accept_counts(self):
        a = (C.c_int64 * 9)()
        self._chk(self.lib.gph_mcmc_accept_counts(self.mcmc, a), "accept_counts")
        return list(a)

    def last_kernel_ms(self, which):
        ms = C.c_double()
        self._chk(self.lib.gph_engine_last_kernel_ms(self.engine, which, C.byref(ms)), "kernel_ms")
        return ms.value

    def class_stats(self, which, reset=False):
        o = (C.c_double * 5)()
        self._chk(self.lib.gph_engine_class_stats(self.engine, which, o, int(reset)), "class_stats")
        return dict(launches=o[0], ms=o[1], evals=o[2], bytes=o[3], nodes=o[4])

    def host_stats(self):
        a, b, c, r = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        self._chk(self.lib.gph_engine_host_stats(self.engine, C.byref(a), C.byref(b), C.byref(c), C.byref(r)), "host_stats")
        return dict(syncs=a.value, collectives=b.value, launches=c.value, resident=bool(r.value))

    def set_timing(self, mask):
        self._chk(self.lib.gph_engine_set_timing(self.engine, mask), "set_timing")

    def unit(self, op, arg=0):
        """kernel-level fixture calls (gph_engine_unit): rows = local loci in input order"""
        stride = max(13, 3 * (self.pack.n - 1), 1 + 5 * (2 * self.pack.n - 1))
        out = np.zeros((self.end - self.begin, stride))
        self._chk(self.lib.gph_engine_unit(self.engine, op, arg, out.ctypes.data_as(C.POINTER(C.c_double)), stride), "unit")
        return out

    def debug_oob(self):
        """(first out-of-range index of a checked build or 0, 1 if the library is a checked build)"""
        w, c = C.c_int32(), C.c_int32()
        self._chk(self.lib.gph_engine_debug_oob(self.engine, C.byref(w), C.byref(c)), "debug_oob")
        return w.value, c.value

    def hbm_bytes(self):
        b = C.c_double()
        self._chk(self.lib.gph_engine_hbm_bytes(self.engine, C.byref(b)), "hbm_bytes")
        return b.value

    def close(self):
        if self.mcmc:
            self.lib.gph_mcmc_destroy(self.mcmc)
            self.mcmc = None
        if self.engine:
            self.lib.gph_engine_destroy(self.engine)
            self.engine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
