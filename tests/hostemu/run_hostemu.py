"""Debugging aid (tests only): run the product's per-locus code compiled for the HOST
(GPH_HOSTEMU, 1-lane wave) on a pack and write the proposal records / state dump, so the
engine's logic can be checked against the oracle in the GPU-less build container.
Not a product path: libgphocs_hip.so contains no CPU code."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gphocs_amd as G  # noqa: E402

HOSTEMU = os.path.join(REPO, "tests", "hostemu", "libgphocs_hostemu.so")


def build_hostemu(sanitize=False, big=False, mid=False, two_walks=False, wave64=False):
    """big: the reference's own caps 200 / 39 / 100 (library variant `n`: node sets of seven words, band lists in LDS) -- a
    separate host build, as the capacities are compile-time.  mid: 64 leaves / 39 populations / 16 bands, the configuration
    of library variants `g` and `h` (GPH_BIG_TREE with two-word node sets, the nibble band list, the fused trace_pair walk
    parking its state over s_targets): ADVICE round 4 -- the only CPU build of that configuration, also under the sanitizers"""
    csrc = os.path.join(REPO, "g-phocs_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in ("gph_engine.hip", "gph_mcmc.cpp", "gph_input.cpp", "gph_program.cpp", "gph_readtrace.cpp", "gph_comm.cpp")]
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] + [os.path.abspath(__file__)]
    out = HOSTEMU.replace(".so", "_h.so") if big else HOSTEMU.replace(".so", "_mid.so") if mid else HOSTEMU
    if two_walks:      # -DGPH_TWO_WALKS: traceLineage(0) and traceLineage(1) as two separate walks (the many-band builds' form)
        out = out.replace(".so", "_2w.so")
    if wave64:
        # -DGPH_EMU64 (round 6): the DEVICE forms of lik_compute / prune_node_q / add_phases / ordered_sum64 / edges_for_time_pop on a
        # 64-lane micro-wave of fibers (csrc/gph_emu64.h) inside the host build: what the sanitizers could not see before
        # (with mid / big: the list-driven device forms of the big-tree builds run on the micro-wave instead of the lane-per-node ones)
        out = out.replace(".so", "_w64.so")
    if sanitize:
        out = out.replace(".so", "_san.so")

    import hashlib
    h = hashlib.sha256()
    for d in sorted(deps):
        h.update(os.path.basename(d).encode() + b"\0" + open(d, "rb").read())
    want = h.hexdigest()[:16]
    side = out + ".buildid"

    def fresh():
        # by content, not by time stamp (checkouts and snapshot copies shuffle those): the id sits next to the library
        try:
            return os.path.exists(out) and open(side).read().strip() == want
        except OSError:
            return False
    if fresh():
        return out
    # several processes may ask at once (the ranks of `bench.py --gpus N --host-emulation`): one builds, into a
    # temporary file that is renamed when complete; the others wait for the lock and find the library up to date
    import fcntl
    with open(out + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if fresh():
                return out
            r = _build(out, srcs, sanitize, big, mid, two_walks, wave64)
            with open(side, "w") as f:
                f.write(want + "\n")
            return r
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(out, srcs, sanitize, big, mid=False, two_walks=False, wave64=False):
    # the engine's hard caps (library variant `x`): every golden fits, the image size does not matter on the host
    caps = ["-DGPH_CAP_LEAVES=200", "-DGPH_CAP_K=40", "-DGPH_CAP_B=100"] if big else \
           ["-DGPH_CAP_LEAVES=64", "-DGPH_CAP_K=40", "-DGPH_CAP_B=16"] if mid else ["-DGPH_CAP_LEAVES=32", "-DGPH_CAP_K=32", "-DGPH_CAP_B=16"]
    tmp = f"{out}.tmp.{os.getpid()}"
    if two_walks:
        caps = caps + ["-DGPH_TWO_WALKS"]
    if wave64:
        caps = caps + ["-DGPH_EMU64"]
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-DGPH_HOSTEMU", "-DGPH_LOGSTEPS"] + caps + [
           "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
           "-x", "c++"] + srcs + ["-lrt", "-o", tmp]
    if sanitize:
        # + the index checks of the checked build (gph_rt.h: GPH_BOUNDS): an index that stays inside the LDS IMAGE but leaves its
        # array is invisible to AddressSanitizer (the image is one struct)
        cmd[1:1] = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-DGPH_BOUNDS"]
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    return out


def run(pack_path, iters, trace, state=None, state_iter=None, with_cond=True, lib=None):
    lib = lib or G.load_library(build_hostemu())
    pk = G.Pack.load(pack_path)
    s = G.Sampler(pk, lib=lib)
    s.set_record_file(trace)
    s.initialize()
    if state and state_iter is not None and state_iter < 0:
        s.dump_state(state, with_cond)
    for it in range(iters):
        s.iteration(it)
        if state and state_iter == it:
            s.dump_state(state, with_cond)
    s.set_record_file(None)
    oob = s.debug_oob()
    s.close()
    assert oob[0] == 0, f"checked build: index out of range at {oob[0]} (line + 100000 x file: 1 gph_locus.h, 2 gph_kernels.h)"


if __name__ == "__main__":
    a = sys.argv
    run(a[1], int(a[2]), a[3], a[4] if len(a) > 4 else None, int(a[5]) if len(a) > 5 else int(a[2]) - 1)
