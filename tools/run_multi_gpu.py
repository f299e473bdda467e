#!/usr/bin/env python3
"""One G-PhoCS chain over several MI355X, one process per GPU:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/run_multi_gpu.py <control-file> [secondary]

Every rank reads the control and sequence files, keeps its contiguous block of loci resident on its own GPU and
runs the same host code; the <= 240-byte reduced vectors of each global proposal are combined with one RCCL
all-gather (torch.distributed, backend nccl) and summed in rank order on every rank.  Rank 0 writes the trace file.
This is gph_run_control_file_ranked() of the library; torch is plumbing for the collective only.
  --backend gloo --lib <path>   CPU form used by the tests (host-emulation build of the engine sources)
"""
import argparse
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ctl")
    ap.add_argument("ctl2", nargs="?")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--lib", default=None)
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import gphocs_amd as G
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = a.backend == "nccl"
    if gpu:
        torch.cuda.set_device(local_rank)
    dist.init_process_group(a.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank) if gpu else torch.device("cpu")
    SLOTS = 64
    hbuf = torch.zeros(SLOTS, dtype=torch.float64)
    dbuf = torch.zeros(SLOTS, dtype=torch.float64, device=dev)
    dout = torch.zeros(world * SLOTS, dtype=torch.float64, device=dev)

    def hook(user, sums, nsum, mins, nmin):
        try:
            s = np.ctypeslib.as_array(sums, shape=(nsum,)) if nsum else np.zeros(0)
            m = np.ctypeslib.as_array(mins, shape=(nmin,)) if nmin else np.zeros(0)
            hb = hbuf.numpy()
            hb[:nsum] = s
            hb[nsum:nsum + nmin] = m
            dbuf.copy_(hbuf)
            dist.all_gather_into_tensor(dout, dbuf)
            rows = dout.cpu().numpy().reshape(world, SLOTS)
            if nsum:
                acc = rows[0, :nsum].copy()
                for r in range(1, world):
                    acc += rows[r, :nsum]          # rank order: the same additions on every rank
                s[:] = acc
            if nmin:
                m[:] = rows[:, nsum:nsum + nmin].min(axis=0)
            return 0
        except Exception as ex:  # pragma: no cover
            print("all-gather hook failed:", ex, file=sys.stderr)
            return 1

    cb = G.ALLREDUCE_FN(hook)
    if a.lib:
        lib = G.load_library(a.lib)
    else:
        # the control file decides the capacity variant (leaves, populations, bands)
        probe = G.load_library()
        ctl = C.c_void_p()
        if probe.gph_control_read(os.fsencode(a.ctl), os.fsencode(a.ctl2) if a.ctl2 else None, C.byref(ctl)):
            sys.exit(1)
        cfg = G.GphConfig()
        probe.gph_control_get(ctl, C.byref(cfg), None, None)
        dims = (cfg.n, cfg.K, cfg.B)
        probe.gph_control_free(ctl)
        lib = G.load_library(dims=dims)
    rc = lib.gph_run_control_file_ranked(os.fsencode(a.ctl), os.fsencode(a.ctl2) if a.ctl2 else None, local_rank if gpu else 0,
                                         int(a.verbose), rank, world, cb, None)
    if rc:
        # a failed rank must not enter another collective: its peers are blocked in the all-gather of the reduction
        # point it never reached, and a barrier here would only wait for the backend's timeout.  Exiting non-zero
        # makes torch.distributed.run tear the job down.
        sys.stderr.write(f"rank {rank}: gph_run_control_file_ranked failed with status {rc}\n")
        os._exit(1)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)


if __name__ == "__main__":
    main()
