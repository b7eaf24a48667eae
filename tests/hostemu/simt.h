// TEST INFRASTRUCTURE: a small SIMT emulator for HIP kernels on the host.
//
// hipLaunchKernelGGL runs one workgroup at a time; every thread of the workgroup is a cooperatively scheduled fiber
// (a six-register x86-64 stack switch; ucontext's per-switch sigprocmask syscall made it 3x slower).  A fiber runs until it finishes or blocks in
//   * __syncthreads()           -- released when every live thread of the workgroup has arrived, or
//   * a wave collective         -- __shfl_xor/_up/__shfl, __ballot, DPP (__builtin_amdgcn_update_dpp), readlane, MFMA:
//                                  the lane deposits its operands and yields; when no lane of its wave is runnable any more,
//                                  the lanes waiting at the same call site are resolved together
//                                  with the others treated as inactive (exec mask off), which is what the hardware does
//                                  for divergent code.
// Wave = 64 consecutive threadIdx.x.  __shared__ variables become statics (one workgroup runs at a time), dynamic LDS is a
// fixed host buffer, LDS atomics are plain operations.  This executes the REAL kernel sources -- including their
// cross-lane reductions and LDS traffic -- so the CPU suite checks their logic; it says nothing about performance and
// it encodes the MFMA / DPP lane layouts as this project uses them (validated on hardware by the GPU suite).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// simt_switch(&save_sp, load_sp): push the callee-saved registers, park the stack pointer, adopt the other stack.
extern "C" void simt_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".weak simt_switch\n"
    ".type simt_switch,@function\n"
    "simt_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size simt_switch, .-simt_switch\n");

// AddressSanitizer build (QTTS_HOSTEMU_ASAN=1 python tests/hostemu/build.py): tell the runtime about every stack switch.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define SIMT_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#endif
#endif

namespace simt {

enum State { RUNNABLE, AT_BARRIER, AT_WAVE_OP, DONE };
enum Op { OP_NONE, OP_SHFL, OP_BALLOT, OP_MFMA_BF16, OP_MFMA_F32 };

struct Fiber {
    void* sp = nullptr;               // parked stack pointer while the fiber is suspended
    void* fake = nullptr;             // ASan fake-stack handle of the suspended fiber
    std::vector<char> stack;
    dim3 tid;
    State state = RUNNABLE;
    // pending wave collective
    Op op = OP_NONE;
    unsigned site = 0;               // call site of the pending collective: lanes waiting at the same site form one group
    uint32_t in32 = 0; int src_lane = 0; int pred = 0;
    uint32_t out32 = 0; uint64_t out64 = 0;
    short a16[8], b16[8]; float a32 = 0, b32 = 0; float c[4], d[4];
};

struct Machine {
    std::vector<Fiber> f;
    void* sched_sp = nullptr;
    void* sched_fake = nullptr; const void* sched_bottom = nullptr; size_t sched_size = 0;     // ASan bookkeeping
    Fiber* cur = nullptr;
    dim3 block_idx, block_dim, grid_dim;
    const std::function<void()>* body = nullptr;
    int order = 0;                    // fiber scheduling order: 0 ascending thread id, 1 descending, >= 2 seeded shuffle of waves
    int block_order = 0;              // workgroup execution order of a launch: 0 ascending blockIdx, 1 descending, >= 2 seeded shuffle
};
inline Machine& M() { static Machine m; return m; }

inline void yield_to_scheduler() {
    Machine& m = M();
    if (m.cur->stack.empty()) {       // launch_sequential: there is no fiber to suspend
        fprintf(stderr, "simt: a kernel compiled as thread-independent (QTTS_SIMT_SEQUENTIAL) called a barrier / cross-lane op\n");
        abort();
    }
#ifdef SIMT_ASAN
    Fiber* self = m.cur;
    __sanitizer_start_switch_fiber(&self->fake, m.sched_bottom, m.sched_size);
#endif
    simt_switch(&m.cur->sp, m.sched_sp);
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(self->fake, &m.sched_bottom, &m.sched_size);
#endif
}
inline void fiber_entry() {
    Machine& m = M();
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &m.sched_bottom, &m.sched_size);
#endif
    (*m.body)();
    m.cur->state = DONE;
#ifdef SIMT_ASAN
    __sanitizer_start_switch_fiber(nullptr, m.sched_bottom, m.sched_size);      // nullptr: this fiber's fake stack dies
#endif
    simt_switch(&m.cur->sp, m.sched_sp);
    abort();                          // a finished fiber is never resumed
}

inline float bf16_bits_to_f32(short h) { uint32_t u = ((uint32_t)(uint16_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// Resolve ONE pending collective of a wave (lanes [w0, w1)): the lanes waiting at the same call site form a group.
// Hardware runs a wave in lockstep and reconverges diverged lanes, so lanes that left a loop early must NOT run ahead and
// complete a later cross-lane op without their partners.  The emulator therefore resolves a single group per pass:
// a group that contains every lane not parked at the workgroup barrier if there is one, otherwise the group at the
// textually earliest call site (the lagging lanes -- still inside the loop / the then-branch -- make progress first and
// meet the others at the later site).
inline bool resolve_wave(Machine& m, unsigned w0, unsigned w1) {
    unsigned best_site = 0; Op best_op = OP_NONE; bool found = false;
    unsigned waiting = 0;
    for (unsigned i = w0; i < w1; ++i) if (m.f[i].state == AT_WAVE_OP) ++waiting;
    if (!waiting) return false;
    for (unsigned i = w0; i < w1; ++i) {
        const Fiber& f = m.f[i];
        if (f.state != AT_WAVE_OP) continue;
        unsigned cnt = 0;
        for (unsigned j = w0; j < w1; ++j) cnt += m.f[j].state == AT_WAVE_OP && m.f[j].op == f.op && m.f[j].site == f.site;
        if (cnt == waiting) { best_site = f.site; best_op = f.op; found = true; break; }      // everybody is here
        if (!found || f.site < best_site) { best_site = f.site; best_op = f.op; found = true; }
    }
    {
        const Op op = best_op; const unsigned site = best_site;
        bool active[64] = {false};
        for (unsigned j = w0; j < w1; ++j) active[j - w0] = m.f[j].state == AT_WAVE_OP && m.f[j].op == op && m.f[j].site == site;
        if (op == OP_SHFL) {
            for (unsigned j = w0; j < w1; ++j) {
                if (!active[j - w0]) continue;
                int s = m.f[j].src_lane;
                const bool ok = s >= 0 && s < (int)(w1 - w0) && active[s];
                m.f[j].out32 = ok ? m.f[w0 + s].in32 : m.f[j].in32;       // inactive / out-of-range source: own value
            }
        } else if (op == OP_BALLOT) {
            uint64_t mask = 0;
            for (unsigned j = w0; j < w1; ++j) if (active[j - w0] && m.f[j].pred) mask |= 1ull << (j - w0);
            for (unsigned j = w0; j < w1; ++j) if (active[j - w0]) m.f[j].out64 = mask;
        } else if (op == OP_MFMA_BF16 || op == OP_MFMA_F32) {
            // lane l: A[m = l%16][k-block l/16], B[k-block l/16][n = l%16]; D[m = 4*(l/16)+r][n = l%16]
            const int KB = op == OP_MFMA_BF16 ? 8 : 1;
            float A[16][32] = {{0}}, B[32][16] = {{0}};
            for (unsigned j = w0; j < w1; ++j) {
                const int l = j - w0;
                if (!active[l]) continue;
                for (int e = 0; e < KB; ++e) {
                    const int k = (l / 16) * KB + e;
                    A[l % 16][k] = op == OP_MFMA_BF16 ? bf16_bits_to_f32(m.f[j].a16[e]) : m.f[j].a32;
                    B[k][l % 16] = op == OP_MFMA_BF16 ? bf16_bits_to_f32(m.f[j].b16[e]) : m.f[j].b32;
                }
            }
            const int K = 4 * KB;
            for (unsigned j = w0; j < w1; ++j) {
                const int l = j - w0;
                if (!active[l]) continue;
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (l / 16) + r, col = l % 16;
                    float acc = m.f[j].c[r];
                    for (int k = 0; k < K; ++k) acc = fmaf(A[row][k], B[k][col], acc);
                    m.f[j].d[r] = acc;
                }
            }
        }
        for (unsigned j = w0; j < w1; ++j) if (active[j - w0]) { m.f[j].state = RUNNABLE; m.f[j].op = OP_NONE; }
    }
    return true;
}

inline void run_block(const std::function<void()>& body, dim3 bidx, dim3 bdim, dim3 gdim) {
    Machine& m = M();
    const unsigned n = bdim.x * bdim.y * bdim.z;
    m.block_idx = bidx; m.block_dim = bdim; m.grid_dim = gdim; m.body = &body;
    if (m.f.size() < n) m.f.resize(n);
    for (unsigned i = 0; i < n; ++i) {
        Fiber& f = m.f[i];
        if (f.stack.empty()) f.stack.resize(256 * 1024);
        f.tid = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
        f.state = RUNNABLE; f.op = OP_NONE; f.site = 0;
        // fresh stack: six zeroed callee-saved slots, then fiber_entry as the `ret` target (rsp = 8 mod 16 on entry)
#ifdef SIMT_ASAN
        __asan_unpoison_memory_region(f.stack.data(), f.stack.size());          // redzones of the previous occupant
#endif
        uintptr_t top = ((uintptr_t)f.stack.data() + f.stack.size()) & ~(uintptr_t)15;
        void** q = (void**)top;
        *--q = nullptr;
        *--q = (void*)&fiber_entry;
        for (int r = 0; r < 6; ++r) *--q = nullptr;
        f.sp = (void*)q;
    }
    // The hardware runs the waves of a workgroup in no particular order.  Results must not depend on it, so the order in
    // which runnable fibers are resumed is selectable (hostemu_set_fiber_order): a kernel that lacks a barrier between an
    // LDS write and a read from another wave passes in one order and fails in another.
    static thread_local std::vector<unsigned> ord;
    ord.resize(n);
    for (unsigned i = 0; i < n; ++i) ord[i] = m.order == 1 ? n - 1 - i : i;
    if (m.order >= 2) {
        const unsigned nw = (n + 63) / 64;
        std::vector<unsigned> wv(nw);
        for (unsigned w = 0; w < nw; ++w) wv[w] = w;
        uint64_t st = 0x9E3779B97F4A7C15ull * (uint64_t)m.order + bidx.x * 0x2545F4914F6CDD1Dull + bidx.y;
        for (unsigned w = nw; w > 1; --w) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(wv[w - 1], wv[(st >> 33) % w]);
        }
        const bool rev = (st >> 7) & 1;
        unsigned k = 0;
        for (unsigned w = 0; w < nw; ++w) {
            const unsigned base = wv[w] * 64, c = n - base < 64 ? n - base : 64;
            for (unsigned l = 0; l < c; ++l) ord[k++] = base + (rev ? c - 1 - l : l);
        }
    }
    for (;;) {
        bool ran = false, live = false;
        for (unsigned oi = 0; oi < n; ++oi) {
            const unsigned i = ord[oi];
            Fiber& f = m.f[i];
            if (f.state == DONE) continue;
            live = true;
            if (f.state != RUNNABLE) continue;
            m.cur = &f;
#ifdef SIMT_ASAN
            __sanitizer_start_switch_fiber(&m.sched_fake, f.stack.data(), f.stack.size());
#endif
            simt_switch(&m.sched_sp, f.sp);
#ifdef SIMT_ASAN
            __sanitizer_finish_switch_fiber(m.sched_fake, nullptr, nullptr);
#endif
            ran = true;
        }
        if (!live) break;
        if (ran) continue;
        // nothing runnable: resolve wave collectives first, then the workgroup barrier
        bool progressed = false;
        for (unsigned w0 = 0; w0 < n; w0 += 64) progressed |= resolve_wave(m, w0, w0 + 64 < n ? w0 + 64 : n);
        if (progressed) continue;
        bool all_at_barrier = true;
        for (unsigned i = 0; i < n; ++i) if (m.f[i].state != DONE && m.f[i].state != AT_BARRIER) all_at_barrier = false;
        if (all_at_barrier) { for (unsigned i = 0; i < n; ++i) if (m.f[i].state == AT_BARRIER) m.f[i].state = RUNNABLE; continue; }
        fprintf(stderr, "simt: deadlock (threads wait at different synchronisation points)\n");
        abort();
    }
}

template <class F>
inline void launch(dim3 grid, dim3 block, F&& body) {
    std::function<void()> fn = body;
    // The hardware starts the workgroups of a launch in no promised order and runs them concurrently; a kernel whose workgroups talk to
    // each other (arrival tickets, last-arriver reductions) must give the same result for every order: hostemu_set_block_order.
    const int bo = M().block_order;
    std::vector<unsigned> xs(grid.x);
    for (unsigned i = 0; i < grid.x; ++i) xs[i] = bo == 1 ? grid.x - 1 - i : i;
    if (bo >= 2) {
        uint64_t st = 0x9E3779B97F4A7C15ull * (uint64_t)bo + grid.x;
        for (unsigned w = grid.x; w > 1; --w) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(xs[w - 1], xs[(st >> 33) % w]);
        }
    }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bi = 0; bi < grid.x; ++bi) run_block(fn, dim3(xs[bi], by, bz), block, grid);
}

// Thread-independent kernels (no barrier, no cross-lane op) do not need fibers: translation units compiled with
// -DQTTS_SIMT_SEQUENTIAL run every thread as a plain call (a blocking primitive then aborts with a message).
template <class F>
inline void launch_sequential(dim3 grid, dim3 block, F&& body) {
    Machine& m = M();
    static Fiber one;
    m.block_dim = block; m.grid_dim = grid; m.cur = &one;
    one.state = DONE;                               // marks "not a fiber": see yield_to_scheduler
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                m.block_idx = dim3(bx, by, bz);
                for (unsigned t = 0; t < block.x; ++t) { one.tid = dim3(t, 0, 0); body(); }
            }
}
// the launch limits of the real device (gfx950): a launch the hardware would reject must not pass here either
inline void check_launch(dim3 grid, dim3 block, size_t shmem, const char* name) {
    const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
    if (threads == 0 || threads > 1024 || grid.x == 0 || grid.y == 0 || grid.z == 0 || grid.x > 2147483647u || grid.y > 65535u ||
        grid.z > 65535u || shmem > 160 * 1024) {
        fprintf(stderr, "simt: launch of %s outside the device limits: grid (%u,%u,%u) block (%u,%u,%u) dynamic LDS %zu\n", name,
                grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
        abort();
    }
}
// Stream capture (hipStreamBeginCapture .. hipStreamEndCapture in hip/hip_runtime.h): while a capture is open, a launch
// is RECORDED -- kernel, geometry and by-value arguments, exactly what a hipGraph kernel node bakes in -- instead of
// executed; hipGraphLaunch replays the recorded nodes in order.  Anything the engine keeps in "device" memory is read at
// replay time, anything it passed by value is frozen at capture time, as on the device.
struct Graph { std::vector<std::function<void()>> nodes; };
struct CaptureState { Graph* open = nullptr; };
inline CaptureState& capture() { static CaptureState c; return c; }

template <class K, class... A>
inline void launch_kernel(dim3 grid, dim3 block, K kern, A... args) {      // arguments are evaluated once, by value, like a real launch
#ifdef QTTS_SIMT_SEQUENTIAL
    auto run = [=] { launch_sequential(grid, block, [=] { kern(args...); }); };
#else
    auto run = [=] { launch(grid, block, [=] { kern(args...); }); };
#endif
    if (capture().open) capture().open->nodes.push_back(run); else run();
}

// ---- what device code calls
inline int lane_id() { const Fiber& f = *M().cur; return (int)(f.tid.x % 64); }     // 1-D workgroups only
inline void barrier() { M().cur->state = AT_BARRIER; yield_to_scheduler(); }
inline uint32_t shfl_bits(uint32_t v, int src, unsigned site) {
    Fiber& f = *M().cur;
    f.op = OP_SHFL; f.site = site; f.in32 = v; f.src_lane = src; f.state = AT_WAVE_OP;
    yield_to_scheduler();
    return f.out32;
}
template <class T> inline T shfl_any(T v, int src, unsigned site) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    uint32_t b; memcpy(&b, &v, 4); b = shfl_bits(b, src, site); T r; memcpy(&r, &b, 4); return r;
}
inline uint64_t ballot(int pred, unsigned site) {
    Fiber& f = *M().cur;
    f.op = OP_BALLOT; f.site = site; f.pred = pred; f.state = AT_WAVE_OP;
    yield_to_scheduler();
    return f.out64;
}
// DPP row rotations / readlane as used by csrc/common.h
inline int update_dpp(int old, int src, int ctrl, unsigned site) {
    (void)old;
    const int l = lane_id();
    if (ctrl >= 0x121 && ctrl <= 0x12F) { const int nrot = ctrl - 0x120; return shfl_any(src, (l & ~15) | ((l + 16 - nrot) & 15), site); }
    fprintf(stderr, "simt: unsupported dpp_ctrl 0x%x\n", ctrl); abort();
}
// MFMA (lane layouts: see resolve_wave)
template <class V8, class V4> inline V4 mfma_bf16(V8 a, V8 b, V4 c, unsigned site) {
    Fiber& f = *M().cur;
    for (int e = 0; e < 8; ++e) { f.a16[e] = a[e]; f.b16[e] = b[e]; }
    for (int r = 0; r < 4; ++r) f.c[r] = c[r];
    f.op = OP_MFMA_BF16; f.site = site; f.state = AT_WAVE_OP;
    yield_to_scheduler();
    V4 d; for (int r = 0; r < 4; ++r) d[r] = f.d[r]; return d;
}
template <class V4> inline V4 mfma_f32(float a, float b, V4 c, unsigned site) {
    Fiber& f = *M().cur;
    f.a32 = a; f.b32 = b;
    for (int r = 0; r < 4; ++r) f.c[r] = c[r];
    f.op = OP_MFMA_F32; f.site = site; f.state = AT_WAVE_OP;
    yield_to_scheduler();
    V4 d; for (int r = 0; r < 4; ++r) d[r] = f.d[r]; return d;
}
// LDS-DMA: lane i's `size` bytes land at dst + i*size (lane-linear destination)
inline void global_load_lds(uintptr_t src, uintptr_t dst, int size) {
    memcpy(reinterpret_cast<char*>(dst) + (size_t)lane_id() * size, reinterpret_cast<const void*>(src), (size_t)size);
}

}  // namespace simt

#define threadIdx (simt::M().cur->tid)
#define blockIdx (simt::M().block_idx)
#define blockDim (simt::M().block_dim)
#define gridDim (simt::M().grid_dim)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    (simt::check_launch((grid), (block), (size_t)(shmem), #kern), simt::launch_kernel((grid), (block), kern, ##__VA_ARGS__))
#define hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, ev_start, ev_stop, flags, ...) \
    ((void)(ev_start), (void)(ev_stop), hipLaunchKernelGGL(kern, grid, block, shmem, stream, ##__VA_ARGS__))
inline void __syncthreads() { simt::barrier(); }
#define __shfl_xor(v, mask) simt::shfl_any((v), simt::lane_id() ^ (mask), __COUNTER__ + 1)
#define __shfl_up(v, delta) simt::shfl_any((v), simt::lane_id() - (delta), __COUNTER__ + 1)
#define __shfl(v, src) simt::shfl_any((v), (src), __COUNTER__ + 1)
#define __ballot(pred) simt::ballot((pred), __COUNTER__ + 1)
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) simt::update_dpp((old), (src), (ctrl), __COUNTER__ + 1)
#define __builtin_amdgcn_wave_barrier() ((void)simt::ballot(1, __COUNTER__ + 1))   /* a wave-level sync point for the emulator: lanes exchange data through LDS around it */
#define __builtin_amdgcn_exp2f(x) exp2f(x)                                  /* v_exp_f32 */
#define __builtin_amdgcn_sinf(x) sinf((x) * 6.283185307179586f)        /* v_sin_f32: input in revolutions */
#define __builtin_amdgcn_readfirstlane(v) (v)     /* only ever applied to wave-uniform values */
#define __builtin_amdgcn_readlane(v, l) simt::shfl_any((v), (l), __COUNTER__ + 1)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) simt::mfma_bf16((a), (b), (c), __COUNTER__ + 1)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) simt::mfma_f32((a), (b), (c), __COUNTER__ + 1)
#define __builtin_amdgcn_global_load_lds(src, dst, size, off, aux) simt::global_load_lds((uintptr_t)(src), (uintptr_t)(dst), (size))
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }
// agent-scope atomics between workgroups (arrival tickets): one workgroup runs at a time, so plain operations
#define __HIP_MEMORY_SCOPE_AGENT 4
namespace simt { template <class T, class V> inline T atomic_fetch_add(T* p, V v) { const T o = *p; *p = (T)(o + v); return o; } }
#define __hip_atomic_fetch_add(p, v, order, scope) simt::atomic_fetch_add((p), (v))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
#define __hip_atomic_load(p, order, scope) (*(p))
inline int atomicMin(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }
