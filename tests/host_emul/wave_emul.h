// wave_emul.h -- TEST INFRASTRUCTURE: a 64-lane wavefront on the host.
//
// The device code of pdmp3_amd/csrc/decode_core.h is written per lane (SIMT).  To run THE SAME code on a CPU the
// 64 lanes of a wave become 64 fibers (own stack, cooperative switching); every cross-lane operation of the device
// (phase fence, lane shuffle, ballot, v_permlane32_swap, v_mfma_f32_16x16x4_f32) is a rendezvous: a fiber deposits
// its operand and yields to the next lane; since all lanes execute the same sequence of rendezvous points (the
// control flow around them is wave-uniform, as it must be on the device), "lane 0 runs again" means all 64 have
// arrived.  The matrix instruction is evaluated with the hardware's fragment layout and arithmetic: lane
// (j = l & 15, kq = l >> 4) supplies A[row j][k = kq] and B[k = kq][col j] and receives D[row 4 kq + r][col j],
// each a k-ordered fmaf chain (cdna_hip_programming.md, "FP32-input MFMA").
#pragma once

#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <functional>

#if !defined(__x86_64__)
#error "wave_emul.h: the fiber switch is written for x86-64 SysV"
#endif

extern "C" void pd_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl pd_ctx_switch
.type pd_ctx_switch,@function
pd_ctx_switch:
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
.size pd_ctx_switch, .-pd_ctx_switch
)");

namespace pdmp3 {
namespace emu {

constexpr int kLanes = 64;
constexpr size_t kStackBytes = 256 * 1024;

struct Wave {
  void* sp[kLanes];
  void* main_sp;
  char* stacks;
  int cur;
  std::function<void()>* body;
  float fa[kLanes], fb[kLanes];
  int ia[kLanes], ib[kLanes];
  bool flag[kLanes];
};

// Several waves (a workgroup) can be live at once: they are run one at a time, a wave hands the processor to the next
// live one where the device code sleeps in a wait loop (wave_yield) and when it ends.
constexpr int kMaxWaves = 16;
struct Group {
  Wave wave[kMaxWaves];
  bool live[kMaxWaves];
  int n, cur;
  void* main_sp;
};
static thread_local Group g_group;
static thread_local Wave* g_cur = &g_group.wave[0];
#define g_wave (*g_cur)

inline int lane() { return g_wave.cur; }
inline int wave_index() { return g_group.cur; }

// lane 0 of the running wave (all its lanes parked in a rendezvous): run the next live wave; returns when it is this wave's turn again
inline void switch_wave(bool ended) {
  Group& G = g_group;
  const int me = G.cur;
  if (ended) G.live[me] = false;
  int nx = -1;
  for (int k = 1; k <= G.n; k++) { const int c = (me + k) % G.n; if (G.live[c]) { nx = c; break; } }
  void* dummy;
  if (nx < 0) { pd_ctx_switch(&dummy, G.main_sp); abort(); }          // the last wave has ended
  if (nx == me) return;
  G.cur = nx;
  g_cur = &G.wave[nx];
  Wave& a = G.wave[me];
  Wave& b = G.wave[nx];
  pd_ctx_switch(ended ? &dummy : &a.sp[a.cur], b.sp[b.cur]);
}

static void fiber_main() {
  Wave& w = g_wave;
  (*w.body)();
  // lanes finish in order: hand over to the next one; the last one passes the processor to the next live wave (or back to the caller)
  const int me = w.cur;
  void* dummy;
  if (me + 1 < kLanes) { w.cur = me + 1; pd_ctx_switch(&dummy, w.sp[me + 1]); }
  else switch_wave(true);
  abort();   // never resumed
}

inline void wave_sync() {
  Wave& w = g_wave;
  const int me = w.cur, nx = (me + 1) % kLanes;
  w.cur = nx;
  pd_ctx_switch(&w.sp[me], w.sp[nx]);
}

// device code sleeping in a wave-uniform wait loop: let the other waves of the workgroup run
inline void wave_yield() {
  wave_sync();
  if (lane() == 0) switch_wave(false);
  wave_sync();
}

static void prepare_wave(Wave& w, std::function<void()>* body) {
  if (!w.stacks) w.stacks = (char*)aligned_alloc(64, kStackBytes * kLanes);
  w.body = body;
  for (int l = 0; l < kLanes; l++) {
    uintptr_t top = (uintptr_t)(w.stacks + kStackBytes * (l + 1));
    top &= ~(uintptr_t)15;
    void** s = (void**)top;
    *--s = nullptr;                    // keeps the entry frame 16-byte aligned (as after a call)
    *--s = (void*)&fiber_main;         // `ret` target of the first switch
    for (int k = 0; k < 6; k++) *--s = nullptr;   // rbp rbx r12 r13 r14 r15
    w.sp[l] = (void*)s;
  }
  w.cur = 0;
}

// run bodies[i] once per lane as wave i, all of them live together (a workgroup)
inline void run_waves(std::function<void()>* bodies, int n) {
  Group& G = g_group;
  if (n > kMaxWaves) abort();
  G.n = n;
  for (int i = 0; i < n; i++) { prepare_wave(G.wave[i], &bodies[i]); G.live[i] = true; }
  G.cur = 0;
  g_cur = &G.wave[0];
  pd_ctx_switch(&G.main_sp, G.wave[0].sp[0]);
  G.cur = 0;
  g_cur = &G.wave[0];
}
// run `body` once per lane as one wave
inline void run_wave(std::function<void()> body) { run_waves(&body, 1); }

inline float shfl_xor(float v, int mask) {
  Wave& w = g_wave;
  const int me = w.cur;
  w.fa[me] = v;
  wave_sync();
  const float r = w.fa[(me ^ mask) & 63];
  wave_sync();
  return r;
}

inline bool any(bool c) {
  Wave& w = g_wave;
  w.flag[w.cur] = c;
  wave_sync();
  bool r = false;
  for (int l = 0; l < kLanes; l++) r = r || w.flag[l];
  wave_sync();
  return r;
}

inline unsigned long long ballot(bool c) {
  Wave& w = g_wave;
  w.flag[w.cur] = c;
  wave_sync();
  unsigned long long r = 0;
  for (int l = 0; l < kLanes; l++) r |= (unsigned long long)(w.flag[l] ? 1 : 0) << l;
  wave_sync();
  return r;
}

// v_permlane32_swap_b32 vdst = a, src = b: lanes 32..63 of a exchange with lanes 0..31 of b
inline void permlane32_swap(int* a, int* b) {
  Wave& w = g_wave;
  const int me = w.cur;
  w.ia[me] = *a;
  w.ib[me] = *b;
  wave_sync();
  if (me < 32) *b = w.ia[me + 32];
  else *a = w.ib[me - 32];
  wave_sync();
}

inline void mfma16(float a, float b, float* cd) {
  Wave& w = g_wave;
  const int me = w.cur, j = me & 15, kq = me >> 4;
  w.fa[me] = a;                         // A[row j][k = kq]
  w.fb[me] = b;                         // B[k = kq][col j]
  wave_sync();
  for (int r = 0; r < 4; r++) {
    const int row = 4 * kq + r;
    float d = cd[r];
    for (int k = 0; k < 4; k++) d = fmaf(w.fa[16 * k + row], w.fb[16 * k + j], d);
    cd[r] = d;
  }
  wave_sync();
}

}  // namespace emu
}  // namespace pdmp3
