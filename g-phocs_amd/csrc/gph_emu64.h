// gph_emu64.h -- TEST BUILDS ONLY (-DGPH_HOSTEMU -DGPH_EMU64; never part of libgphocs_hip*.so).
//
// A 64-lane wavefront for the HOST build of the engine sources, so that the DEVICE forms of the lane-parallel functions --
// lik_compute (ballot fix-point, lane per node / lane per pattern), prune_node_q, add_phases, ordered_sum64,
// edges_for_time_pop -- compile for the host and run the goldens under AddressSanitizer / UBSan in the GPU-less container
// (VERDICT round 5, item 7).  The rest of the per-locus code keeps its one-lane host form; when it reaches one of those
// functions the dispatcher (gph_locus.h) starts a MICRO-WAVE: 64 fibers run the device form on the same LDS image, and every
// cross-lane primitive (ballot, lane read, crossbar broadcast, DPP shift, barrier / wave fence) and every store to the image
// that goes through an accessor is a RENDEZVOUS: all 64 lanes must arrive at the same site before any goes on -- the lockstep
// of a wavefront at the granularity the code can observe.  A store site is two rendezvous (everybody has computed its value;
// everybody has stored): a uniform read-modify-write like setCNT(k, CNT(k) + 1), executed by all lanes, counts once.
// Lanes that wait at different sites (divergence the device forms do not have) or a lane that ends while others wait abort
// the run with the two sites named.
//
// Fibers: 64 stacks per host thread, a hand-written x86-64 context switch (callee-saved registers + stack pointer: the switch
// of a rendezvous costs tens of nanoseconds, a glibc swapcontext is a system call); annotated for AddressSanitizer.
#pragma once
#if defined(GPH_HOSTEMU) && defined(GPH_EMU64)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define GPH_EMU_ASAN 1
#else
#define GPH_EMU_ASAN 0
#endif
#if !defined(__x86_64__)
#error "gph_emu64.h: the fiber switch is written for x86-64"
#endif

extern "C" void gph_emu_switch(void **save_sp, void *new_sp);

namespace gph_emu {
constexpr int W = 64;
constexpr size_t STACK = 512 * 1024;
struct Wave {
  bool active = false;
  int cur = 0, arrived = 0, ndone = 0;
  long gen = 0;
  void *sp[W] = {nullptr};
  void *sched_sp = nullptr;
  char *stack[W] = {nullptr};
  long wait_gen[W];
  bool done[W];
  int site[W];
  uint64_t xbuf[2][W];
  const std::function<void()> *fn = nullptr;
  long rendezvous_total = 0, waves_total = 0;
};
extern thread_local Wave tw;
extern int g_enabled;              // run the device forms as micro-waves (default 1; GPH_EMU64=0 in the environment: host forms)

inline bool in_wave() { return tw.active; }
inline int lane() { return tw.active ? tw.cur : 0; }
inline int nlanes() { return tw.active ? W : 1; }
inline bool enabled() { return g_enabled != 0 && !tw.active; }

void to_scheduler();
void run(const std::function<void()> &fn);

// all 64 lanes arrive at `site` before any goes on; outside a micro-wave: nothing
inline void rendezvous(int site)
{
  Wave &w = tw;
  if (!w.active) return;
  const int me = w.cur;
  const long g = w.gen;
  w.site[me] = site;
  w.rendezvous_total++;
  if (++w.arrived == W) {
    for (int l = 0; l < W; l++)
      if (w.site[l] != site) {
        fprintf(stderr, "gph_emu64: lanes diverged: lane %d waits at site %d, lane %d at site %d (source line, or 8xx: a store through an accessor)\n",
                me, site, l, w.site[l]);
        abort();
      }
    w.arrived = 0;
    w.gen++;
    return;
  }
  w.wait_gen[me] = g;
  while (w.gen == g) to_scheduler();
  w.wait_gen[me] = -1;
}
// every lane hands in a value, every lane sees all 64 (two buffers by generation parity: a lane can be one rendezvous ahead)
inline const uint64_t *exchange(uint64_t v, int site)
{
  Wave &w = tw;
  const int par = (int)(w.gen & 1);
  w.xbuf[par][w.cur] = v;
  rendezvous(site);
  return w.xbuf[par];
}
inline uint64_t ballot(bool p, int site)
{
  if (!tw.active) return p ? 1 : 0;
  const uint64_t *b = exchange(p ? 1 : 0, site);
  uint64_t m = 0;
  for (int l = 0; l < W; l++) m |= (b[l] & 1) << l;
  return m;
}
inline int readlane32(int v, int l, int site)
{
  if (!tw.active) return v;
  return (int)(uint32_t)exchange((uint32_t)v, site)[l & 63];
}
inline double readlane64(double v, int l, int site)
{
  if (!tw.active) return v;
  uint64_t u;
  memcpy(&u, &v, 8);
  u = exchange(u, site)[l & 63];
  double r;
  memcpy(&r, &u, 8);
  return r;
}
// v_mov_b32_dpp wave_shl:1 with bound_ctrl: lane i takes lane i + 1, the last lane 0 (tools/probe/dpp_probe.cpp has the direction)
inline int dpp_wave_shl1(int v, int site)
{
  if (!tw.active) return 0;
  const int me = tw.cur;
  const uint64_t *b = exchange((uint32_t)v, site);
  return me + 1 < W ? (int)(uint32_t)b[me + 1] : 0;
}
// ds_bpermute_b32: lane i takes the value of lane (addr_i >> 2) & 63
inline int bpermute(int addr, int v, int site)
{
  if (!tw.active) return v;
  return (int)(uint32_t)exchange((uint32_t)v, site)[(addr >> 2) & 63];
}
}   // namespace gph_emu

#ifdef GPH_EMU64_IMPL
asm(R"(
.text
.globl gph_emu_switch
.type gph_emu_switch,@function
gph_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size gph_emu_switch,.-gph_emu_switch
)");
namespace gph_emu {
thread_local Wave tw;
int g_enabled = [] { const char *e = getenv("GPH_EMU64"); return e ? atoi(e) : 1; }();
static void fiber_switch(void **save, void *to, const void *to_stack_bottom, size_t to_size)
{
#if GPH_EMU_ASAN
  void *fake = nullptr;
  __sanitizer_start_switch_fiber(&fake, to_stack_bottom, to_size);
  gph_emu_switch(save, to);
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
  (void)to_stack_bottom; (void)to_size;
  gph_emu_switch(save, to);
#endif
}
static thread_local const void *sched_stack_bottom = nullptr;
static thread_local size_t sched_stack_size = 0;
void to_scheduler()
{
  Wave &w = tw;
  fiber_switch(&w.sp[w.cur], w.sched_sp, sched_stack_bottom, sched_stack_size);
}
static void fiber_entry()
{
#if GPH_EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &sched_stack_bottom, &sched_stack_size);
#endif
  Wave &w = tw;
  (*w.fn)();
  Wave &w2 = tw;
  w2.done[w2.cur] = true;
  w2.ndone++;
  /* a finished lane never runs again */
#if GPH_EMU_ASAN
  __sanitizer_start_switch_fiber(nullptr, sched_stack_bottom, sched_stack_size);
#endif
  gph_emu_switch(&w2.sp[w2.cur], w2.sched_sp);
  abort();
}
void run(const std::function<void()> &fn)
{
  Wave &w = tw;
  if (w.active) { fprintf(stderr, "gph_emu64: a micro-wave inside a micro-wave\n"); abort(); }
  for (int l = 0; l < W; l++) {
    if (!w.stack[l]) {
      void *p = nullptr;
      if (posix_memalign(&p, 64, STACK)) abort();
      w.stack[l] = (char *)p;
    }
    uint64_t *top = (uint64_t *)(w.stack[l] + STACK);
    *--top = 0;                               /* keeps the entry function's frame 16-byte aligned */
    *--top = (uint64_t)(uintptr_t)&fiber_entry;
    for (int k = 0; k < 6; k++) *--top = 0;   /* rbp rbx r12 r13 r14 r15 */
    w.sp[l] = top;
    w.done[l] = false;
    w.wait_gen[l] = -1;
    w.site[l] = 0;
  }
  w.fn = &fn;
  w.arrived = 0; w.ndone = 0;
  w.active = true;
  w.waves_total++;
  while (w.ndone < W) {
    bool progressed = false;
    for (int l = 0; l < W; l++) {
      if (w.done[l] || (w.wait_gen[l] >= 0 && w.wait_gen[l] == w.gen)) continue;
      w.cur = l;
      progressed = true;
      fiber_switch(&w.sched_sp, w.sp[l], w.stack[l], STACK);
    }
    if (!progressed) {
      int a = -1, b = -1;
      for (int l = 0; l < W; l++) { if (w.done[l]) a = l; else b = l; }
      fprintf(stderr, "gph_emu64: deadlock: %d lanes wait (lane %d at site %d) while %d have left the function (lane %d)\n",
              W - w.ndone, b, b >= 0 ? w.site[b] : 0, w.ndone, a);
      abort();
    }
  }
  w.active = false;
  w.fn = nullptr;
}
}   // namespace gph_emu
#endif   /* GPH_EMU64_IMPL */
#endif
