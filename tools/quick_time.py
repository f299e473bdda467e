#!/usr/bin/env python3
"""Quick timing of the HIP path on a pack file: iterations/s, evals/s, per-kernel ms."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G  # noqa: E402


def main():
    pack = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    t0 = time.time()
    pk = G.Pack.load(pack)
    print(f"pack loaded in {time.time()-t0:.1f}s: L={pk.L} n={pk.n} K={pk.K} B={pk.B}", flush=True)
    s = G.Sampler(pk)
    t0 = time.time()
    s.initialize()
    print(f"init {time.time()-t0:.3f}s  init kernel {s.last_kernel_ms(3):.3f} ms  hbm {s.hbm_bytes()/1e6:.1f} MB", flush=True)
    for it in range(warm):
        s.iteration(it)
    s.counters(reset=True)
    t0 = time.time()
    ks = {0: 0.0, 1: 0.0, 2: 0.0}
    for it in range(warm, warm + iters):
        s.iteration(it)
        for k in ks:
            ks[k] += s.last_kernel_ms(k)
    dt = time.time() - t0
    c = s.counters()
    print(f"{iters} iterations in {dt:.3f}s -> {iters/dt:.2f} it/s, {c['evals']/dt/1e6:.3f} M evals/s, "
          f"R/eval {c['eval_nodes']/max(c['evals'],1):.2f}, bytes/eval {c['eval_bytes']/max(c['evals'],1):.0f}")
    print(f"per-iteration kernel ms: sweep {ks[0]/iters:.3f}  last tau_eval {ks[1]/iters:.3f}  mix_eval {ks[2]/iters:.3f}")
    print("accept counts", s.accept_counts(), "state", {k: (v if isinstance(v, float) else None) for k, v in s.state().items()})
    s.close()


if __name__ == "__main__":
    main()
