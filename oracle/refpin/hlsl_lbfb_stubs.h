// ORACLE pin (test infrastructure only): stand-ins for what Rtxpt/Lighting/LightsBaker.hlsl binds, so that the text of its NEE-AT feedback passes
// (ProcessFeedbackHistoryPreFilter / P0 / P1a / P1b / P2 (FillTile) / P3, ClearFeedbackHistory and the helpers they call) compiles as C++ and runs thread by thread.
// Included inside namespace hl::lbfb by hlsl_tu.py --integrator, after LightingTypes.hlsli (LightingControlData, LightFeedbackReservoir) and MicroRng.hlsli.
//   * resources: the pointer-carrying stubs of hlsl_pt_stubs.h; out-of-range texture writes are dropped and reads return 0, as D3D does for UAVs
//   * the dispatch: every thread of a group is a coroutine (ucontext); GroupMemoryBarrierWithGroupSync yields to the scheduler, which resumes the group's threads
//     only when all of them have arrived — group-shared memory and barriers behave as on the GPU, whatever the kernel does between them
//   * P0's counters: the TARGET_VULKAN variant of the text (one InterlockedAdd per thread; the D3D12 variant merges equal lanes with wave intrinsics first: same sums)
//   * NEEAT_ENABLE_DEBUG_DRAW 0, LLB_ENABLE_VALIDATION 0; the LLB_* sizes are NEEATBaker.hlsli:15-22
#include <ucontext.h>
#include <functional>
#define NEEAT_ENABLE_DEBUG_DRAW 0
#define LLB_ENABLE_VALIDATION 0
#define TARGET_VULKAN 1
#define LLB_NUM_COMPUTE_THREADS 128
#define LLB_NUM_COMPUTE_THREADS_2D 8
#define LLB_PREPROCESS_BLOCK_SIZE_OUTER 16
#define LLB_PREPROCESS_BLOCK_SIZE_INNER (LLB_PREPROCESS_BLOCK_SIZE_OUTER-2)
#define groupshared static
static LightingControlData g_ctrl;
struct PinControlBuffer { LightingControlData& operator[](uint) const { return g_ctrl; } };
static PinControlBuffer u_controlBuffer;
static LightsBakerConstants g_bakerConstants;      // (LightingControlData carries it as padding unless NEEAT_BAKER_ONLY is set; the path tracer's TU, where this is compiled, leaves it unset)
#define g_bakerConsts g_bakerConstants
#define g_controlInfo u_controlBuffer[0]
static RWTexture2D<float> u_feedbackTotalWeight, u_feedbackTotalWeightScratch, u_feedbackTotalWeightBlended, u_historyDepth;
static RWTexture2D<uint> u_feedbackCandidates, u_feedbackCandidatesScratch, u_feedbackCandidatesBlended;
static RWBuffer<uint> u_perLightProxyCounters, u_lightSamplingProxies, u_localSamplingBuffer, u_historyRemapPastToCurrent, u_historyRemapCurrentToPast; static RWBuffer<float> u_lightWeights; static RWBuffer<uint> u_scratchList;
struct PinMotion { const uint* p = nullptr; uint w = 0;      // ScreenMotionVectors as the build pass stored them (RGBA16F, two words per pixel); null: zero — reference mode (Sample.cpp:2494)
    float3 operator[](int2 c) const { if (!p) return float3(0, 0, 0); const uint* q = p + 2 * ((size_t)c.y * w + (size_t)c.x); return float3(ptref::f16tof32(q[0] & 0xffffu), ptref::f16tof32(q[0] >> 16), ptref::f16tof32(q[1] & 0xffffu)); } };
static RWTexture2D<float> t_depthBuffer; static PinMotion t_motionVectors;      // reference mode: the depth the path tracer exported last frame, zero motion; realtime mode: the build pass's depth and motion vectors of this frame
static inline void InterlockedAdd(uint& dst, uint v) { dst += v; }
// integer vector helpers the passes use (HLSL: clamp on int2; int against uint compares as uint)
using hl::clamp;
static inline int2 clamp(int2 v, int2 lo, int2 hi) { return int2(v.x < lo.x ? lo.x : (v.x > hi.x ? hi.x : v.x), v.y < lo.y ? lo.y : (v.y > hi.y ? hi.y : v.y)); }
#define LBFB_MIXED_CMP(op) static inline bool2 operator op(int2 a, uint2 b) { return bool2((uint)a.x op b.x, (uint)a.y op b.y); } static inline bool2 operator op(uint2 a, int2 b) { return bool2(a.x op (uint)b.x, a.y op (uint)b.y); }
LBFB_MIXED_CMP(<) LBFB_MIXED_CMP(>=)
#undef LBFB_MIXED_CMP

// ---- a thread group as coroutines
struct GroupRunner {
    std::vector<ucontext_t> ctx; std::vector<std::vector<char> > stacks; std::vector<char> done; ucontext_t sched; uint cur = 0;
    std::function<void(uint)> body; std::function<void(uint)> atBarrier;
};
static GroupRunner* g_group = nullptr;
static void GroupMemoryBarrierWithGroupSync() { GroupRunner* g = g_group; swapcontext(&g->ctx[g->cur], &g->sched); }
static void AllMemoryBarrierWithGroupSync() { GroupMemoryBarrierWithGroupSync(); }
static void group_trampoline() { GroupRunner* g = g_group; g->body(g->cur); g->done[g->cur] = 1; swapcontext(&g->ctx[g->cur], &g->sched); }
// runs body(thread) for thread = 0 .. n-1 as one group; atBarrier(k) is called once every thread has reached its k-th barrier (k = 1, 2, ...)
static void run_group(uint n, std::function<void(uint)> body, std::function<void(uint)> atBarrier = nullptr) {
    GroupRunner g; g.ctx.resize(n); g.stacks.assign(n, std::vector<char>(128 * 1024)); g.done.assign(n, 0); g.body = body; g.atBarrier = atBarrier;
    g_group = &g;
    for (uint i = 0; i < n; i++) { getcontext(&g.ctx[i]); g.ctx[i].uc_stack.ss_sp = g.stacks[i].data(); g.ctx[i].uc_stack.ss_size = g.stacks[i].size(); g.ctx[i].uc_link = nullptr; makecontext(&g.ctx[i], group_trampoline, 0); }
    for (uint barrier = 1;; barrier++) {
        bool alive = false;
        for (uint i = 0; i < n; i++) if (!g.done[i]) { g.cur = i; swapcontext(&g.sched, &g.ctx[i]); alive = alive || !g.done[i]; }
        if (!alive) break;
        if (atBarrier) atBarrier(barrier);
    }
    g_group = nullptr;
}
