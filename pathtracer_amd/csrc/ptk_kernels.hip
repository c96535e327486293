// HIP kernels for gfx950 and the C ABI of include/ptk.h.
//
// One lane per path.  A one-wave workgroup renders an 8x8 pixel tile (or one chunk of its
// samples): the tile's (pixel, sample) pairs are a pool of work items that the wave's 64 lanes
// draw from, so the primary rays of a refill are neighbours and no lane idles while another
// still has samples to do.  The traversal stack lives in LDS, a lane's entries next to each
// other (stack[lane][level], see LdsStack); the tile's pixel sums live in LDS too.  Workgroup ids are remapped so that the workgroups that share an
// XCD (and its L2) render the chunks of one tile, then the neighbouring tile, and so that every
// XCD gets an equal share of every part of the image.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/ptk.h"
#include "ptk_build.h"
#include "ptk_lbvh.h"
#include "ptk_shade.h"

namespace ptk {

constexpr int kBlock = 64;  // one wave per workgroup: a finished wave frees its slot at once
constexpr int kTile = 8;    // a workgroup renders an 8x8 pixel tile

// Wave scheduling knobs of the render kernel (see its main loop).  Measured on the 81,920-triangle
// mesh scene (1024x1024, 1024 spp, 3 bounces), each against the values below: descend 8: -8%,
// 12: -2%, 20: -1%, 24: -3%; lit 4: same, 16: -1%; shade 16: -1%, 32: -1%; item 4: -1%, 16: -3%;
// 4 waves per SIMD (109 VGPRs, none spilled): -8%.  Swept again for the 6-wave form: the same values stay best.
#ifndef PTK_DESCEND_MIN
#define PTK_DESCEND_MIN 20  // leave the descend loop when fewer lanes than this are at an inner node (measured again with the
                            // rotated trees: 20 against 16 is +0.2 % / +1.1 % on the two mesh scenes, 12 and 24 lose)
#endif
#ifndef PTK_DESCEND_MIN_TINY
#define PTK_DESCEND_MIN_TINY 16  // ... in the whole-scene-in-LDS form (20: -0.7 % on the Cornell box), and in the ray-batch kernel
#endif
#ifndef PTK_LIT_MIN
#define PTK_LIT_MIN 8       // consume finished shadow rays once this many lanes hold one
#endif
#ifndef PTK_SHADE_MIN
#define PTK_SHADE_MIN 24    // shade once this many lanes hold a finished path segment
#endif
#ifndef PTK_ITEM_MIN
#define PTK_ITEM_MIN 8      // hand out work items once this many lanes are free
#endif
#ifndef PTK_ITEM_MIN_TINY
#define PTK_ITEM_MIN_TINY 16  // ... in the whole-scene-in-LDS form, where a refill (camera ray with its lens rejection loop) is a larger share of the work: +1.3 % on the Cornell box
#endif
#ifndef PTK_MIN_WAVES
#define PTK_MIN_WAVES 5     // __launch_bounds__ waves per SIMD of the render kernel's deep-tree form: 96 VGPRs
#endif
#ifndef PTK_MIN_WAVES_SHALLOW
#define PTK_MIN_WAVES_SHALLOW 6  // ... and of its shallow-tree form: 80 VGPRs (3 spilled), 6400 bytes of LDS
#endif
// A frame whose tiles alone give the chip only a few rounds of waves ends with most CUs idle
// while the slowest tiles finish, so a tile's samples are split over several workgroups.
// 1024x1024, 1024 spp: 1 chunk 4.9k, 2: 6.1k, 4: 6.9k, 8: 7.1k, 16: 6.9k Mrays/s (before the
// work-item pool; with it the optimum is flat between 8 and 32).
#ifndef PTK_CHUNK_ROUNDS
#define PTK_CHUNK_ROUNDS 32       // split tiles into sample chunks until the grid is this many times the resident waves
#endif
#ifndef PTK_CHUNK_MIN_SAMPLES
#define PTK_CHUNK_MIN_SAMPLES 16  // ... but never below this many samples per chunk
#endif
constexpr int kMaxChunks = 32;

// Device memory of ptk_render_static, the render entry point that allocates nothing: the per-chunk partial sums
// of a frame go here instead of into a buffer that grows on demand.  One copy per GPU, at its own address
// (ptk_create asks the runtime for it with hipGetSymbolAddress: in host code the name `scratch` is not a
// device address).  Every call on a GPU shares that GPU's copy.
constexpr size_t kStaticScratchBytes = (size_t)128 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
constexpr int kCounterSlots = 24;  // 0-4 the ptk_stats counters, 8-14 wave-phase counters and 16-20 phase clocks of a PTK_PHASE_STATS build, 22 the ray-batch kernel's work counter
constexpr int kIntersectQueueSlot = 22;
constexpr size_t kUnitQueueBytes = 8 * 32 * sizeof(unsigned int);  // eight counters, 128 bytes apart
constexpr int kIntersectRunMax = 1024;  // most rays a wave draws from the work counter at a time

// A build with -DPTK_PHASE_STATS makes instrumented launches also count, per wave, how often each
// phase of the render loop ran and how many lanes it ran for (printed by ptk_get_stats when the
// environment has PTK_PHASE_STATS set).  A profiling aid only; normal builds compile it away.
#ifdef PTK_PHASE_STATS
#define PTK_PHASE(...) __VA_ARGS__
#else
#define PTK_PHASE(...)
#endif
constexpr int kDescendMin = PTK_DESCEND_MIN;
// Wave priority per phase (s_setprio: which wave the SIMD's arbiter serves first when several have an instruction
// ready).  Waves in the dependent load -> test -> stack chain of a traversal go before waves that shade or take new
// work: +0.9 % on the 82k-triangle scene, +2.4 % on the Cornell box, +2.6 % on the 1.3M-triangle scene against no
// priorities; the reverse order loses as much, and the ray-batch kernel (no shading to yield to) loses 2 % with them.
#ifndef PTK_PRIO_DESCEND
#define PTK_PRIO_DESCEND 3
#endif
#ifndef PTK_PRIO_LEAF
#define PTK_PRIO_LEAF 2
#endif
#ifndef PTK_PRIO_SHADE
#define PTK_PRIO_SHADE 0
#endif

// Layout of the per-lane stacks in the workgroup's LDS array.  Lane-major (a lane's entries are consecutive
// words, lanes kStackDepth words apart - an odd number, so lanes at the same level still fall into different
// banks): moving the top is +-4 bytes, which fits an instruction's inline constant; level-major
// (stack[level][lane]) needs +-256 and two registers to hold those.
#ifndef PTK_STEAL_UNITS
#define PTK_STEAL_UNITS 1  // persistent render launches: a workgroup whose XCD has no units left takes other XCDs' units
#endif
// 1: every XCD renders one contiguous block of tile rows instead of every eighth row (experiment: each L2 then sees
// one part of the image; measured 1-2 % slower on every scene, the 1.3M-triangle one included: the even mix of cheap
// and costly image regions per XCD that it gives up is worth more than the locality it buys).
#ifndef PTK_ROWS_COMPACT
#define PTK_ROWS_COMPACT 0
#endif
#ifndef PTK_STACK_LANE_MAJOR
#define PTK_STACK_LANE_MAJOR 1
#endif
// 1: the phases of the render loop only store their lanes' new queries; the arithmetic that starts a query (three
// reciprocals, the slab constants, the stack) runs once per pass of the loop for all of them (trav_start).
#ifndef PTK_DEFER_START
#define PTK_DEFER_START 1
#endif
// n > 1: a step of the descend loop visits up to n nodes per lane before the wave looks at its lanes again
#ifndef PTK_DESCEND_VISITS
#define PTK_DESCEND_VISITS 2
#endif
#ifndef PTK_STACK_PREFETCH
#define PTK_STACK_PREFETCH 1  // a node visit reads its stack's top entry before it asks for the node (0: where the visit ends)
#endif
#ifndef PTK_STACK_ROOM
#define PTK_STACK_ROOM 8  // steps that the descend loop runs unchecked after one look at every lane's stack height
#endif
// OVER > 0: the stack may grow OVER entries beyond its DEPTH entries of LDS.  Those entries live in the lane's
// private memory (`over`, a few words of scratch).  A traversal stack holds one entry per level at which BOTH
// children were hit and the far one still waits, so even in a 25-level tree more than nineteen pending entries is a
// rarity.  Every operation of this type first asks, for the whole wave at once (one compare and a scalar branch),
// whether any lane is that far up, and only then takes the path that looks at each entry's level.  The descend
// loop does better: roomy() tells it (one ballot) that no lane is within kRoom entries of the end of its LDS
// part, and it then runs kRoom steps on unchecked(), the same stack seen as a plain LDS stack whose operations
// check nothing.  This is what lets deep trees run with the LDS footprint - and so the resident waves - of
// shallow ones.
template <int DEPTH, int OVER = 0>
struct LdsStack {
  static constexpr int kDepth = DEPTH;
  static constexpr int kStep = PTK_STACK_LANE_MAJOR ? 1 : kBlock;  // words between a lane's consecutive entries
  static_assert(OVER == 0 || PTK_STACK_LANE_MAJOR, "the overflow entries assume lane-major stacks");
  // Pointers in the LDS address space (32 bits, ds_read / ds_write), stated explicitly: once the overflow path
  // selects between an LDS and a private address the compiler no longer works that out by itself and falls back to
  // 64-bit generic pointers and flat loads for every stack access.
  using Word = PTK_LDS_SPACE uint32_t;
  Word* top;   // this lane's top entry (beyond base + DEPTH - 1: an address that is never touched, only counted with)
  Word* base;  // this lane's entry 0
  volatile uint32_t* over;  // entries DEPTH, DEPTH + 1, ...: an array of OVER words in the lane's private memory (its own
                            // object, so that top and base stay in registers; volatile, so that it stays memory and is
                            // not turned into OVER more live registers)
  static constexpr int kOver = OVER > 0 ? OVER : 1;
  static constexpr int kRoom = PTK_STACK_ROOM;
  static_assert(OVER == 0 || DEPTH - 2 - kRoom >= 1, "a stack with overflow entries needs room for the unchecked steps");
  using Unchecked = LdsStack<DEPTH, 0>;
  __device__ __forceinline__ void attach(uint32_t* array, int lane, volatile uint32_t* over_entries = nullptr)
  {
    base = top = (Word*)array + (PTK_STACK_LANE_MAJOR ? lane * DEPTH : lane);
    over = over_entries;
  }
  // True when every lane of the wave can take kRoom more pushes (one per step, plus the scratch entry above the top)
  // inside its LDS part.  To be called with all lanes of the wave active.
  __device__ __forceinline__ bool roomy() const { return OVER == 0 || __ballot(top > base + (DEPTH - 2 - kRoom)) == 0ull; }
  __device__ __forceinline__ Unchecked unchecked() const
  {
    Unchecked u;
    u.top = top; u.base = base; u.over = nullptr;
    return u;
  }
  __device__ __forceinline__ void reset() { top = base; *top = (uint32_t)kTravDone; }
  // entry at address p, wherever it lives
  __device__ __forceinline__ uint32_t read_at(const Word* p) const
  {
    const int level = (int)(p - base);
    uint32_t v;
    if (level < DEPTH) v = *p;
    else v = over[min(level - DEPTH, OVER - 1)];
    return v;
  }
  __device__ __forceinline__ void write_at(Word* p, uint32_t v)
  {
    const int level = (int)(p - base);
    if (level < DEPTH) *p = v;
    else over[min(level - DEPTH, OVER - 1)] = v;
  }
  // true when some lane's entry at `p` or above may lie outside LDS
  __device__ __forceinline__ bool any_beyond(const Word* p) const
  {
    return OVER > 0 && __builtin_expect(__ballot(p >= base + DEPTH) != 0ull, 0);
  }
  __device__ __forceinline__ void push(uint32_t v)
  {
    top += kStep;
    if (any_beyond(top)) write_at(top, v);
    else *top = v;
  }
  __device__ __forceinline__ uint32_t pop()
  {
    const uint32_t v = any_beyond(top) ? read_at(top) : *top;
    top -= kStep;
    return v;
  }
  // Branch-free on the usual path: the entry above the top is scratch, so v is always written there and only
  // counts if the top moves up.  The builder keeps trees shallower than DEPTH + OVER - 2 (ptk_upload_scene checks).
  // Start of a node visit: the top entry, which step() hands back at the end of the visit.  Read before the node's own
  // loads are issued, it has long arrived when they have.
  __device__ __forceinline__ uint32_t prefetch() const { return PTK_STACK_PREFETCH ? (any_beyond(top) ? read_at(top) : *top) : 0u; }
  __device__ __forceinline__ uint32_t step(bool do_push, uint32_t v, bool do_pop, uint32_t t)
  {
    if (!PTK_STACK_PREFETCH) t = any_beyond(top) ? read_at(top) : *top;
    if (any_beyond(top + kStep)) write_at(top + kStep, v);
    else top[kStep] = v;
    // at most one of the two is set; a byte step of +-4 (lane-major) is an inline constant of the select
    const int step_words = do_push ? kStep : (do_pop ? -kStep : 0);
    top += step_words;
    return t;
  }
};

// The two vectors of a path that only wait while its shadow ray is traced (ParkInState in ptk_shade.h), in LDS: word k
// of lane l at base[k * 64 + l].  Kept in the path state they are live through every traversal loop, and at six
// waves per SIMD that is six registers the compiler otherwise finds in scratch memory (7 spilled dwords -> 2).  Measured:
// the 1536 bytes have to come out of the traversal stacks (13 LDS entries per lane instead of 19), and that costs more
// than the spills did: -0.7 % on the 82k-triangle scene, -1.2 % on the 1.3M-triangle scene, -0.2 % on the glass scene
// (-1.7 %, -3.1 %, -1.3 % since the build options brought the spills down to three).
// Hence off; -DPTK_PARK=1 -DPTK_STACK_DEPTH_TABLES=13 -DPTK_STACK_DEPTH_SHALLOW=16 -DPTK_STACK_ROOM=4 builds it.
// (-DPTK_PARK=2 parks the light term alone, 768 bytes: -DPTK_PARK=2 -DPTK_STACK_DEPTH_TABLES=16 -DPTK_STACK_DEPTH_SHALLOW=19.)
struct LdsPark {
  PTK_LDS_SPACE float* at;  // this lane's word 0
  __device__ __forceinline__ void store(PathState& ps, v3 pending, v3 next_d) const
  {
    at[0] = pending.x; at[kBlock] = pending.y; at[2 * kBlock] = pending.z;
    if (PTK_PARK == 2) { ps.next_d = next_d; return; }
    at[3 * kBlock] = next_d.x; at[4 * kBlock] = next_d.y; at[5 * kBlock] = next_d.z;
  }
  __device__ __forceinline__ void load(const PathState& ps, v3& pending, v3& next_d) const
  {
    pending = V3(at[0], at[kBlock], at[2 * kBlock]);
    next_d = PTK_PARK == 2 ? ps.next_d : V3(at[3 * kBlock], at[4 * kBlock], at[5 * kBlock]);
  }
};

// Per-pixel radiance sums of a tile, in LDS.  Two lanes can finish paths of the same pixel in the
// same step, hence the atomic add (ds_add_f32).
struct LdsAccum {
  float* base;
  __device__ __forceinline__ void add(uint32_t slot, v3 L)
  {
    __hip_atomic_fetch_add(base + slot, L.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(base + 64 + slot, L.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(base + 128 + slot, L.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
};

struct RenderArgs {
  int32_t samples, max_bounce;
  uint32_t seed;
  int32_t sample_first, sample_stride;
  int32_t row_begin, row_end;
  int32_t tiles_x, n_tiles;
  int32_t chunks;  // work units per tile; each takes every chunks-th sample
  int32_t persistent;  // 1: the launch has fewer workgroups than work units, and a workgroup draws further units from unit_queue
  // Rows of the launch -> rows of the image.  A launch covers `rows` rows (a multiple of the tile
  // height, the image edge is cut off by row_end): either rows [row_begin, row_end) of the image, or
  // the interleaved bands band_first, band_first + band_stride, ... of band_rows rows each.
  int32_t rows;
  int32_t band_rows, band_first, band_stride;  // band_rows == 0: no bands
};

__host__ __device__ __forceinline__ int launch_row_to_image_row(const RenderArgs& ra, int r)
{
  if (ra.band_rows == 0) return ra.row_begin + r;
  int j = r / ra.band_rows;
  return (ra.band_first + j * ra.band_stride) * ra.band_rows + (r - j * ra.band_rows);
}

// Workgroups are dealt to the 8 XCDs round-robin, so ids b and b+8 share an L2.  Give each
// residue class a contiguous run of tiles.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int n)
{
  int xcd = b & 7, k = b >> 3;
  int q = n >> 3, r = n & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// Position v in the list 0, 8, 16, ..., 1, 9, 17, ..., 7, 15, ... of the rows below n -> that row.
__device__ __forceinline__ int interleave_rows(int v, int n)
{
  int q = n >> 3, r = n & 7;
  int cut = r * (q + 1);
  if (v < cut) return (v % (q + 1)) * 8 + v / (q + 1);
  v -= cut;
  return (v % q) * 8 + r + v / q;
}

// Sum of a per-lane 32-bit count over the wave, in 64 bits (64 lanes of up to 2^32 - 1 each).
__device__ __forceinline__ unsigned long long wave_sum(uint32_t v)
{
  unsigned long long s = v;
  for (int off = 32; off > 0; off >>= 1) s += (unsigned long long)__shfl_xor((long long)s, off, 64);
  return s;
}

__device__ __forceinline__ void flush_counters(const Counters& c, unsigned long long* out)
{
  unsigned long long a = wave_sum(c.camera_rays), b = wave_sum(c.segments), q = wave_sum(c.queries);
  unsigned long long x = wave_sum(c.box_tests), p = wave_sum(c.prim_tests);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], a);
    atomicAdd(&out[1], b);
    atomicAdd(&out[2], q);
    atomicAdd(&out[3], x);
    atomicAdd(&out[4], p);
  }
}

// The kernel's forms (see ptk_layout.h: stack size, LDS, waves per SIMD): kFormTiny for scenes whose whole tree and
// primitive records fit a few KB of LDS (traversed from there); kFormShallowTables / kFormSpillTables (trees of at
// most / more than kTablesTreeDepth levels) for scenes whose material and light tables fit beside a shorter stack
// in LDS; otherwise kFormShallow for trees of at most kShallowTreeDepth levels and kFormSpill for deeper ones
// (kFormDeep: the older all-LDS stack for those, kept for comparison).
constexpr int kFormShallow = 0, kFormDeep = 1, kFormTiny = 2, kFormSpill = 3, kFormShallowTables = 4, kFormSpillTables = 5;
template <bool STATS, int FORM>
__global__ __launch_bounds__(kBlock, FORM == kFormDeep ? PTK_MIN_WAVES : PTK_MIN_WAVES_SHALLOW) void render_kernel(
    SceneView sc, const DeviceConsts* __restrict__ consts_global, RenderArgs ra, f4* __restrict__ accum, f4* __restrict__ partial,
    unsigned long long* counters, unsigned int* unit_queue)
{
  constexpr bool DEEP = FORM == kFormDeep, TINY = FORM == kFormTiny;
  constexpr bool TABLES = FORM == kFormShallowTables || FORM == kFormSpillTables;  // material and light tables in LDS
  constexpr bool SPILL = FORM == kFormSpill || FORM == kFormSpillTables || (DEEP && kStackDepthDeep < kStackDepth);  // stack entries beyond the LDS part
  constexpr int kLdsDepth = DEEP ? kStackDepthDeep : (TINY ? kStackDepthTiny : (TABLES ? kStackDepthTables : kStackDepthShallow));
  using LdsStack = ptk::LdsStack<kLdsDepth, SPILL ? kStackDepth - kLdsDepth : 0>;
  using Mem = typename std::conditional<TINY, LdsMem, typename std::conditional<TABLES, TablesMem, GlobalMem>::type>::type;
  constexpr bool DEFER = PTK_DEFER_START != 0;
  constexpr bool PARK = PTK_PARK && !DEEP && !TINY;  // the waiting vectors of a path in LDS (LdsPark)
  static_assert(!TABLES || kStackDepthTables > (PARK ? (PTK_PARK == 2 ? 16 : 13) : 19) || (kStackDepthTables + 3) * kBlock * 4 + kTableBytes + (PARK ? kParkBytes : 0) <= 5 * 1280,
                "stack, pixel sums, tables (and parked vectors) share five LDS granules");
  using Park = typename std::conditional<PARK, LdsPark, ParkInState>::type;
  constexpr int kSceneChunks = (kTinyNodes * sizeof(NodePair) + kTinyPrims * (sizeof(PrimRec) + sizeof(ShadeRec)) +
                                kTinyMaterials * sizeof(MaterialRec) + kTinyLights * sizeof(LightRec)) / sizeof(f4);
  __shared__ f4 lds_scene[TINY ? kSceneChunks : (TABLES ? kTableBytes / sizeof(f4) : 1)];
  Mem mem;
  if constexpr (TINY) {
    // the workgroup's own copy of the scene, all of it but the texels (the __syncthreads below covers it)
    f4* at = lds_scene;
    auto take = [&](const void* from, int records, int chunks_per_record, int capacity) {
      f4* to = at;
      for (int i = (int)threadIdx.x; i < records * chunks_per_record; i += kBlock) to[i] = reinterpret_cast<const f4*>(from)[i];
      at += capacity * chunks_per_record;
      return to;
    };
    mem.nodes = (const PTK_LDS_SPACE char*)take(sc.nodes, sc.n_nodes, sizeof(NodePair) / sizeof(f4), kTinyNodes);
    mem.prims = (const PTK_LDS_SPACE char*)take(sc.prims, sc.n_prims, sizeof(PrimRec) / sizeof(f4), kTinyPrims);
    mem.shades = (const PTK_LDS_SPACE ShadeRec*)take(sc.shade, sc.n_prims, sizeof(ShadeRec) / sizeof(f4), kTinyPrims);
    mem.materials = (const PTK_LDS_SPACE MaterialRec*)take(sc.materials, sc.n_materials, sizeof(MaterialRec) / sizeof(f4), kTinyMaterials);
    mem.lights = (const PTK_LDS_SPACE LightRec*)take(sc.lights, sc.n_lights, sizeof(LightRec) / sizeof(f4), kTinyLights);
  }
  if constexpr (TABLES) {
    // the workgroup's own copy of the light table and, behind it, the material table (ptk_upload_scene checked
    // that both fit; the __syncthreads below covers the copy)
    constexpr int kM = sizeof(MaterialRec) / sizeof(f4), kL = sizeof(LightRec) / sizeof(f4);
    const int light_chunks = sc.n_lights * kL, material_chunks = sc.n_materials * kM;
    for (int i = (int)threadIdx.x; i < light_chunks; i += kBlock) lds_scene[i] = reinterpret_cast<const f4*>(sc.lights)[i];
    for (int i = (int)threadIdx.x; i < material_chunks; i += kBlock) lds_scene[light_chunks + i] = reinterpret_cast<const f4*>(sc.materials)[i];
    mem.lights = (const PTK_LDS_SPACE LightRec*)lds_scene;
    mem.materials = (const PTK_LDS_SPACE MaterialRec*)(lds_scene + light_chunks);
  }
  // read-only for the whole launch: the constant address space lets every use be a scalar load at that use
  const auto* consts = (const __attribute__((address_space(4))) DeviceConsts*)consts_global;
  const CameraView& cam = *(const CameraView*)&consts->cam;
  const FlagView& fl = *(const FlagView*)&consts->fl;
  __shared__ uint32_t lds_stack[LdsStack::kDepth * kBlock];
  __shared__ float lds_sum[3 * kBlock];  // r, g, b sums of the tile's 64 pixel slots
  __shared__ float lds_park[PARK ? kParkBytes / sizeof(float) : 1];
  Park park;
  if constexpr (PARK) park.at = (PTK_LDS_SPACE float*)lds_park + threadIdx.x;
  LdsStack st;
  volatile uint32_t stack_over[LdsStack::kOver];
  st.attach(lds_stack, (int)threadIdx.x, stack_over);

  // Virtual id -> (tile, sample chunk).  An XCD's consecutive workgroups are the chunks of one
  // tile, then the next tile of the same tile row; tile rows are dealt to the XCDs round-robin
  // (row r -> XCD r % 8) so that every XCD sees the same mix of cheap and expensive image regions.
  // A work unit is one (tile, chunk) pair.  With one workgroup per unit the hardware hands them out in order, one
  // XCD after the other, and a unit that waits for a free slot on its XCD holds up the units behind it on the
  // others.  A persistent launch has as many workgroups as fit the GPU at once; each starts on the unit of its own
  // number and then draws the next units of ITS XCD's share from that XCD's counter (unit_queue, one cache line
  // each, zeroed before the launch).  Which workgroup renders a unit does not change what the unit computes or
  // where it writes.
  const int lane = threadIdx.x;
  const int n_units = ra.n_tiles * ra.chunks;
  int xcd = (int)blockIdx.x & 7;  // the XCD whose share the current unit is from (the wave's own, until that runs out)
  auto units_of = [&](int x) { return (n_units >> 3) + (x < (n_units & 7) ? 1 : 0); };                 // an XCD's share
  auto groups_of = [&](int x) { return ((int)gridDim.x >> 3) + (x < ((int)gridDim.x & 7) ? 1 : 0); };  // its workgroups of the launch
  int units_here = units_of(xcd);
  Counters cnt = {0, 0, 0, 0, 0};
  constexpr bool kWaveCounts = !DEEP;
  unsigned long long wave_segments = 0, wave_queries = 0;
  unsigned long long wave_boxes = 0, wave_prims = 0;  // STATS: the lanes' 32-bit test counts, folded in after every unit
  for (int unit = (int)blockIdx.x >> 3; unit < units_here;) {
  int virt = xcd_remap(xcd + 8 * unit, n_units);
  int tile = virt / ra.chunks, chunk = virt - tile * ra.chunks;
  int tx = tile % ra.tiles_x, ty = tile / ra.tiles_x;
#if !PTK_ROWS_COMPACT
  ty = interleave_rows(ty, ra.n_tiles / ra.tiles_x);
#endif

  PathParams pp;
  pp.x0 = tx * kTile; pp.y0 = launch_row_to_image_row(ra, ty * kTile);
  pp.slot_bits = 6; pp.row_bits = 3;
  pp.x_end = cam.width; pp.y_end = ra.row_end;
  pp.max_bounce = ra.max_bounce; pp.seed = ra.seed; pp.camera_shift = consts->camera_shift;
  pp.samples = (ra.samples - chunk + ra.chunks - 1) / ra.chunks;  // chunk c takes samples c, c + chunks, ...
  pp.sample_first = ra.sample_first + chunk * ra.sample_stride;
  pp.sample_stride = ra.sample_stride * ra.chunks;
  LdsAccum acc;
  acc.base = lds_sum;
  lds_sum[lane] = 0.0f; lds_sum[64 + lane] = 0.0f; lds_sum[128 + lane] = 0.0f;
  __syncthreads();
  PathState ps;
  path_init(ps);
  uint32_t next_item = 0;  // wave-uniform: only ever changed where all 64 lanes are active

  // Hand the next work items, in lane order, to the lanes that ask for one.  Returns whether the lane is still free
  // afterwards: its item needed no query (a pixel slot outside the image) - it then asks again in the next refill
  // phase rather than in a loop here, which would cost every refill the register copies of a loop around the whole
  // path state for the sake of the tiles on the image's edge.
  // Path segments and queries started by this wave, counted for the whole wave at once where its phases end
  // (a ballot and a scalar add) instead of in a register of every lane: the per-lane counts of `cnt` are left
  // unread in this kernel, so the compiler drops them.  Camera rays are counted by the host (rows x width x samples).
  // (The deep-tree form has the registers and is 1 % faster with the per-lane counts, so it keeps them.)
  bool started = false;  // DEFER: this lane was given a query in this pass of the loop and trav_start() has not run for it yet
  auto refill = [&](bool need) -> bool {
    const unsigned long long m = __ballot(need);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t item = next_item + rank;
    next_item += (uint32_t)__popcll(m);
    bool still = false;
    if (need) still = path_begin_item<LdsStack, DEFER>(ps, st, sc, cam, pp, item, cnt);
    if (kWaveCounts) {
      const unsigned long long began = (unsigned long long)__popcll(__ballot(need && !still && ps.alive));  // a primary ray is on its way
      wave_segments += began;
      wave_queries += began;
    }
    if (DEFER) started = started || (need && !still && ps.alive);
    return still;
  };
  bool need = refill(true);  // this lane's path has ended (or never began) and it has not been given a new work item yet
  if (DEFER) { if (started) trav_start(ps.tr, st, sc); started = false; }

  // Every lane keeps one query in flight.  The wave alternates uniform phases, each run only when
  // enough lanes want it (or when no lane is traversing any more, so that everything drains):
  //   descend  - lanes at an inner node visit it, repeated while at least kDescendMin lanes are at one
  //              (or while no lane holds a leaf);
  //   leaves   - lanes intersect the primitives of the leaves they hold or stand at;
  //   lit      - lanes whose shadow ray has finished add the light and start the path's next segment;
  //   shade    - lanes whose path segment has finished shade the hit and start a shadow ray or the
  //              next segment;
  //   refill   - lanes whose path has ended take the tile's next work items.
  // Lanes not in the running phase keep their state (node in a register, stack in LDS).
  PTK_PHASE(uint32_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};)
  // clock cycles the wave spends in each phase (s_memtime; the stamps themselves cost about a tenth of the run)
  PTK_PHASE(unsigned long long tph[5] = {0, 0, 0, 0, 0}; unsigned long long tmark = __builtin_amdgcn_s_memtime();)
#define PTK_PHASE_CLOCK(k) PTK_PHASE({ unsigned long long tnow = __builtin_amdgcn_s_memtime(); tph[k] += tnow - tmark; tmark = tnow; })
  __builtin_amdgcn_s_setprio(PTK_PRIO_DESCEND);
  int32_t held = kTravDone;  // a leaf the lane has reached but not intersected yet (kTravDone: none)
  for (;;) {
    // A lane that reaches a leaf puts it aside and goes on descending with the others; it stops only at a
    // second leaf.  The leaf phase then intersects the leaf put aside and the one the lane stands at.
    // (Descending past a leaf uses a closest-hit distance that is not up to date, which costs a few extra
    // node visits and never a wrong answer.  Holding one leaf: +13 % frame rate; holding two: -6 %.)
    // One step of the descend loop on stack `s` (the lane's stack, or its unchecked view); true: leave the loop.
    auto descend_step = [&](auto& s) -> bool {
      using S = typename std::remove_reference<decltype(s)>::type;
      if (trav_at_leaf(ps.tr) && held == kTravDone) { held = ps.tr.node; ps.tr.node = (int32_t)s.pop(); }
      const bool want = trav_at_inner(ps.tr);
      const int n_want = __popcll(__ballot(want));
      // few lanes at an inner node: they wait for the leaf phase to send more lanes back, unless no leaf is held
      const bool any_held = __ballot(held != kTravDone) != 0ull;
      // Evaluated eagerly: one branch instead of three.  (The compiler forms "n_want < kDescendMin" with a 64-bit VECTOR
      // compare; the same test spelled out as two scalar instructions makes the loop 1 % slower: a wave issues its
      // instructions one after the other whatever unit they are for, and the scalar chain of a step is the longer one.)
      if ((n_want == 0) | ((n_want < (TINY ? PTK_DESCEND_MIN_TINY : kDescendMin)) & any_held)) return true;
      PTK_PHASE(ph[0]++; ph[1] += n_want;)
      if (want) trav_inner<S, STATS, Mem>(ps.tr, s, sc, cnt, mem);
#pragma unroll
      for (int extra = 1; extra < PTK_DESCEND_VISITS; extra++)
        if (trav_at_inner(ps.tr)) trav_inner<S, STATS, Mem>(ps.tr, s, sc, cnt, mem);
      return false;
    };
    PTK_MARK("descend");
    for (;;) {
      if (SPILL && st.roomy()) {
        auto fast = st.unchecked();
        bool leave = false;
#pragma nounroll
        for (int k = 0; k < LdsStack::kRoom / PTK_DESCEND_VISITS && !leave; k++) leave = descend_step(fast);  // at most kRoom pushes
        st.top = fast.top;
        if (leave) break;
      } else if (descend_step(st)) {
        break;
      }
    }
    PTK_PHASE_CLOCK(0)
    PTK_MARK("leaves");
    __builtin_amdgcn_s_setprio(PTK_PRIO_LEAF);
    PTK_PHASE(ph[2]++; ph[3] += __popcll(__ballot(held != kTravDone)); ph[4] += __popcll(__ballot(trav_at_inner(ps.tr)));)
    if (held != kTravDone) {
      if (trav_leaf_prims<STATS, Mem>(ps.tr, held, sc, cnt, mem)) ps.tr.node = kTravDone;
      held = kTravDone;
    }
    if (trav_at_leaf(ps.tr)) trav_leaf<LdsStack, STATS, Mem>(ps.tr, st, sc, cnt, mem);
    PTK_PHASE_CLOCK(1)
    PTK_MARK("decide");
    __builtin_amdgcn_s_setprio(PTK_PRIO_SHADE);
    const bool done = trav_done(ps.tr);
    const bool waiting = ps.alive && done && !need;
    const bool lit = waiting && ps.shadow, shade = waiting && !ps.shadow;
    const int n_lit = __popcll(__ballot(lit)), n_shade = __popcll(__ballot(shade));
    const bool idle = __ballot(!done) == 0ull;  // no lane is traversing: everything that waits must run now
    if (idle && n_lit + n_shade + __popcll(__ballot(need)) == 0) break;
    if (n_lit >= PTK_LIT_MIN || (idle && n_lit > 0)) {
      PTK_MARK("lit");
      bool goes_on = false;
      if (lit) {
        goes_on = !path_consume_shadow<LdsStack, Park, DEFER>(ps, st, sc, fl, pp, cnt, park);
        if (!goes_on) { acc.add(ps.slot(), ps.L); need = true; }
      }
      if (DEFER) started = started || goes_on;
      if (kWaveCounts) {
        const unsigned long long began = (unsigned long long)__popcll(__ballot(goes_on));  // the path's next segment
        wave_segments += began;
        wave_queries += began;
      }
    }
    PTK_MARK("decide2");
    PTK_PHASE_CLOCK(2)
    if (n_shade >= PTK_SHADE_MIN || (idle && n_shade > 0)) {
      PTK_PHASE(ph[5]++; ph[6] += n_shade;)
      PTK_MARK("shade");
      bool goes_on = false;
      if (shade) {
        goes_on = !path_shade<LdsStack, Mem, Park, DEFER>(ps, st, sc, fl, pp, cnt, mem, park);
        if (!goes_on) { acc.add(ps.slot(), ps.L); need = true; }
      }
      if (DEFER) started = started || goes_on;
      if (kWaveCounts) {
        wave_queries += (unsigned long long)__popcll(__ballot(goes_on));                 // a shadow ray or the next segment
        wave_segments += (unsigned long long)__popcll(__ballot(goes_on && !ps.shadow));  // the next segment
      }
    }
    PTK_MARK("decide3");
    PTK_PHASE_CLOCK(3)
    const int n_need = __popcll(__ballot(need));
    if (n_need >= (TINY ? PTK_ITEM_MIN_TINY : PTK_ITEM_MIN) || (idle && n_need > 0)) {
      PTK_MARK("refill");
      need = refill(need);
      PTK_MARK("refill_end");
    }
    if (DEFER) { if (started) trav_start(ps.tr, st, sc); started = false; }
    PTK_PHASE_CLOCK(4)
    __builtin_amdgcn_s_setprio(PTK_PRIO_DESCEND);
  }
  PTK_PHASE(ph[7] = (uint32_t)wave_sum(cnt.leaf_iterations); cnt.leaf_iterations = 0;)
  if (STATS) {  // per unit, so that a lane's 32-bit counts cannot wrap however many units a persistent workgroup draws
    wave_boxes += wave_sum(cnt.box_tests);
    wave_prims += wave_sum(cnt.prim_tests);
    cnt.box_tests = cnt.prim_tests = 0;
  }
  if (!kWaveCounts) {
    wave_segments += wave_sum(cnt.segments);
    wave_queries += wave_sum(cnt.queries);
    cnt.segments = cnt.queries = 0;
  }
  PTK_PHASE(if (STATS && (threadIdx.x & 63) == 0) {
    for (int k = 0; k < 8; k++) atomicAdd(&counters[8 + k], (unsigned long long)ph[k]);
    for (int k = 0; k < 5; k++) atomicAdd(&counters[16 + k], tph[k]);
  })
  __syncthreads();
  const int px = pp.x0 + (lane & 7), py = pp.y0 + (lane >> 3);  // lane l writes out pixel slot l
  if (px < pp.x_end && py < pp.y_end) {
    f4 a;
    a.x = lds_sum[lane]; a.y = lds_sum[64 + lane]; a.z = lds_sum[128 + lane]; a.w = (float)pp.samples;
    if (ra.chunks == 1) {
      size_t i = (size_t)py * (size_t)cam.width + (size_t)px;
      f4 b = accum[i];
      b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
      accum[i] = b;
    } else {
      // partial sums are indexed by the launch's own rows
      size_t r = (size_t)(ty * kTile + (lane >> 3));
      size_t i = ((size_t)chunk * (size_t)ra.rows + r) * (size_t)cam.width + (size_t)px;
      partial[i] = a;
    }
  }
  if (TINY || !ra.persistent) break;  // (the tiny form is never launched persistent: its code is better without the loop)
  // the next unit of this XCD's share; when that has run out, of the next XCD's that has units left (the shares are
  // equal in units, not in work: the XCDs with the cheaper tiles would otherwise stand idle at the end of a launch)
  for (int tries = 0; tries < 8; tries++) {
    unsigned int drawn = 0u;
    if (lane == 0) drawn = atomicAdd(unit_queue + xcd * 32, 1u);
    unit = groups_of(xcd) + (int)__builtin_amdgcn_readfirstlane(drawn);
    units_here = units_of(xcd);
    if (unit < units_here || !PTK_STEAL_UNITS) break;
    xcd = (xcd + 1) & 7;
  }
  __syncthreads();  // the pixel sums are zeroed again at the top
  }
  {
    const unsigned long long boxes = wave_boxes, prims = wave_prims;
    if (lane == 0) {
      atomicAdd(&counters[1], wave_segments);
      atomicAdd(&counters[2], wave_queries);
      if (STATS) { atomicAdd(&counters[3], boxes); atomicAdd(&counters[4], prims); }
    }
  }
}

// accum[rows of the launch] += partial[0] + partial[1] + ... in that order (one fixed summation
// order, so a frame does not depend on which workgroup finished first).
__global__ __launch_bounds__(256) void fold_partials_kernel(f4* __restrict__ accum, const f4* __restrict__ partial,
                                                            RenderArgs ra, int width)
{
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t count = (size_t)ra.rows * (size_t)width;
  if (i >= count) return;
  int r = (int)(i / (size_t)width), x = (int)(i - (size_t)r * (size_t)width);
  int y = launch_row_to_image_row(ra, r);
  if (y >= ra.row_end) return;
  size_t o = (size_t)y * (size_t)width + (size_t)x;
  f4 a = accum[o];
  for (int c = 0; c < ra.chunks; c++) {
    f4 p = partial[(size_t)c * count + i];
    a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
  }
  accum[o] = a;
}

// Experiment hooks of the ray-batch kernel: PTK_INTERSECT_WAVES=n asks the compiler for n waves per SIMD,
// PTK_INTERSECT_LDS_PAD=bytes takes LDS away (fewer resident waves), PTK_EXP_NO_HIT_OUTPUT drops the result stores.
#ifndef PTK_INTERSECT_WAVES
#define PTK_INTERSECT_WAVES 1
#endif
template <bool STATS, bool DEEP>
__global__ __launch_bounds__(kBlock, PTK_INTERSECT_WAVES) void intersect_kernel(SceneView sc, const float* __restrict__ rays, int n,
                                                           ptk_hit* __restrict__ hits, unsigned long long* counters, int run)
{
  // (a test build with a very short shallow stack keeps the full one here: a lane's stack region also stages its hit record)
  constexpr int kLdsDepth = (DEEP || kStackDepthShallow < 16) ? kStackDepthDeep : kStackDepthShallow;
  using LdsStack = ptk::LdsStack<kLdsDepth, DEEP ? kStackDepth - kLdsDepth : 0>;  // deep trees: overflow words, every operation checked
  __shared__ uint32_t lds_stack[LdsStack::kDepth * kBlock];
  __shared__ uint32_t lds_finished[kBlock];  // the lanes whose records wait to be written out, in rank order
#ifdef PTK_INTERSECT_LDS_PAD
  __shared__ uint32_t lds_pad[PTK_INTERSECT_LDS_PAD / 4];
  if (n < 0) lds_pad[threadIdx.x] = 1u;  // never true: keeps the array
#endif
  LdsStack st;
  volatile uint32_t stack_over[LdsStack::kOver];
  st.attach(lds_stack, (int)threadIdx.x, stack_over);
  const uint32_t lane = threadIdx.x;
  Counters cnt = {0, 0, 0, 0, 0};
  // The launch holds one round of resident waves (launch_intersect): every further round of workgroups was
  // measured to cost as much as the first.  A wave starts on the run of `run` consecutive rays of its
  // own number, then draws further runs from a counter in device memory until the batch is used up, and works
  // through a run like the render kernel works through a tile: lanes take the next rays as their queries
  // finish, and the wave alternates descend / leaf / hand-over phases.  Drawing (rather than owning a fixed
  // share) keeps all waves reading rays and writing records near one moving front, which the address
  // translation likes (+35 % on 64 M rays), and evens out costly and cheap regions.  Draws from one address
  // complete at about 70 M/s, so runs are as long as the batch allows (launch_intersect); shorter runs towards the end of a batch (tried: a quarter
  // and a sixteenth over the last two draws per wave) cost more in draws than they save in waiting.
  unsigned long long* const queue = counters + kIntersectQueueSlot;
  long long next = 0, end = 0;  // wave-uniform: the run in hand
  bool more = true;             // wave-uniform: the runs drawn so far do not reach the end of the batch
  bool first = true;
  Traversal tr;
  tr.node = kTravDone;
  tr.o = tr.d = V3(0.f, 0.f, 0.f);
  int32_t mine = -1;  // the ray this lane is working on
  float shift = 0.0f; // how far its origin was advanced towards the scene (scene_entry_shift)
  int32_t held = kTravDone;
  bool alive = true;
  // Hit records leave through LDS.  A 52-byte record written by its own lane is thirteen 4-byte stores to
  // sixty-four scattered places per wave; instead a finished lane puts its record into its own (now empty)
  // stack region, and the whole wave then copies the finished records out with consecutive lanes writing
  // consecutive words, so that a store instruction covers whole records.
  constexpr uint32_t kRecordWords = sizeof(ptk_hit) / 4;
  static_assert(sizeof(ptk_hit) % 4 == 0 && kRecordWords + 2 <= (uint32_t)LdsStack::kDepth && PTK_STACK_LANE_MAJOR,
                "a hit record and its ray index must fit a lane's stack region above the sentinel");
  for (;;) {
    const bool done = trav_done(tr);
    const bool need = alive && done;
    const unsigned long long need_mask = __ballot(need);
    const bool idle = __ballot(!done) == 0ull;
    if (idle && need_mask == 0ull) break;
    if (__popcll(need_mask) >= PTK_ITEM_MIN || idle) {
      const bool finished = need && mine >= 0;
      const unsigned long long fin_mask = __ballot(finished);
      if (finished) {
        ptk_hit out;
        memset(&out, 0, sizeof out);
        if (tr.hit.id != kNoPrim) {
          Surface s;
          make_surface(sc, tr.o, tr.d, tr.hit, s);
          out.hit = 1; out.id = tr.hit.id; out.t = tr.hit.t + shift;
          out.point[0] = s.point.x; out.point[1] = s.point.y; out.point[2] = s.point.z;
          out.normal[0] = s.normal.x; out.normal[1] = s.normal.y; out.normal[2] = s.normal.z;
          out.uv[0] = s.uv[0]; out.uv[1] = s.uv[1];
          out.material = (int32_t)s.material; out.inside = s.inside ? 1 : 0;
        }
        auto* rec = st.base + 1;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&out);
#pragma unroll
        for (uint32_t k = 0; k < kRecordWords; k++) rec[k] = w[k];
        rec[kRecordWords] = (uint32_t)mine;
        const uint32_t frank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fin_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fin_mask, 0u));
        lds_finished[frank] = lane;
      }
      if (fin_mask != 0ull) {  // wave-uniform
        __syncthreads();  // one wave: orders the LDS writes above before the reads below
        const uint32_t total = (uint32_t)__popcll(fin_mask) * kRecordWords;
        uint32_t* out_words = reinterpret_cast<uint32_t*>(hits);
        for (uint32_t i = lane; i < total; i += kBlock) {
          const uint32_t r = i / kRecordWords, k = i - r * kRecordWords;
          const uint32_t* rec = lds_stack + lds_finished[r] * (uint32_t)LdsStack::kDepth + 1;
#ifndef PTK_EXP_NO_HIT_OUTPUT
          out_words[(size_t)rec[kRecordWords] * kRecordWords + k] = rec[k];
#else
          if (rec[k] == 0xdeadbeefu) out_words[0] = 1u;
#endif
        }
        __syncthreads();  // the stack regions are reused by the queries that start below
      }
      if (next >= end && more) {  // wave-uniform
        // a wave's first run is the one of its own number: the counter hands out the runs after those
        unsigned long long got = 0ull;
        if (!first && lane == 0) got = atomicAdd(queue, 1ull);
        const long long i = first ? (long long)blockIdx.x
                                  : (long long)gridDim.x + (long long)__builtin_amdgcn_readfirstlane((uint32_t)got);  // fewer than 2^31 / 64 draws
        first = false;
        next = i * run;
        end = next + run < (long long)n ? next + run : (long long)n;
        more = next + run < (long long)n;
      }
      // a lane that finds the run used up keeps asking (the next pass draws a new run) until the batch is used up
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
      const long long take = next + rank;
      next += __popcll(need_mask);
      if (need) {
        mine = -1;
        if (take < end) {
          // origin and direction as three 8-byte loads (a ray is 24 bytes, 8-byte aligned)
          const float2* r = reinterpret_cast<const float2*>(rays) + 3 * (size_t)take;
          const float2 a = r[0], b = r[1], c = r[2];
          mine = (int32_t)take;
          cnt.queries++;
          const v3 o = V3(a.x, a.y, b.x), d = V3(b.y, c.x, c.y);
          shift = scene_entry_shift(sc, o, d);
          trav_begin(tr, st, sc, o + d * shift, d, kTMax, kNoPrim);
        } else if (!more) {
          alive = false;
        }
      }
    }
    for (;;) {  // as in render_kernel: a leaf is held while the lane keeps descending
      if (trav_at_leaf(tr) && held == kTravDone) { held = tr.node; tr.node = (int32_t)st.pop(); }
      const bool want = trav_at_inner(tr);
      const int n_want = __popcll(__ballot(want));
      const bool any_held = __ballot(held != kTravDone) != 0ull;
      if ((n_want == 0) | ((n_want < PTK_DESCEND_MIN_TINY) & any_held)) break;  // evaluated eagerly: one branch instead of three
      if (want) trav_inner<LdsStack, STATS>(tr, st, sc, cnt);
    }
    if (held != kTravDone) {
      if (trav_leaf_prims<STATS>(tr, held, sc, cnt)) tr.node = kTravDone;
      held = kTravDone;
    }
    if (trav_at_leaf(tr)) trav_leaf<LdsStack, STATS>(tr, st, sc, cnt);
  }
  // The waves of a batch end together, and counter updates to one address complete one after another at about
  // 14 ns each: five per wave were 0.4 ms of a 1.5 ms batch.  Without statistics the host counts the rays.
  if (STATS) flush_counters(cnt, counters);
}

// What Renderer::save_image does to every pixel before the file is written (src/renderer.cpp:174-192):
// mean radiance -> ACES curve -> gamma 1/2.2 -> clamp to [0,1] -> * 255, truncated.  Double
// arithmetic as there; the mean is formed exactly as ptk_read_accum forms it on the host (fp32
// reciprocal and product, obtained here through double so that the fast fp32 division of this
// build cannot change a bit).  Only the W*H*3 bytes then cross PCIe instead of W*H*16.
__global__ __launch_bounds__(256) void display_kernel(const f4* __restrict__ accum, uint8_t* __restrict__ out, size_t n,
                                                      unsigned int* max_w_bits)
{
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float w = 0.f;
  if (i < n) {
    const f4 a = accum[i];
    w = a.w;
    const float inv = w > 0.f ? (float)(1.0 / (double)w) : 0.f;
    const float mean[3] = {(float)((double)a.x * (double)inv), (float)((double)a.y * (double)inv),
                           (float)((double)a.z * (double)inv)};
    for (int k = 0; k < 3; k++) {
      double c = (double)mean[k];
      c = (c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14);  // Narkowicz ACES fit, src/renderer.cpp:174-178
      c = pow(c, 1.0 / 2.2);                                         // gamma_correction, src/util.h:131-134
      double q = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);                // map_pixel, src/renderer.cpp:10
      if (!(c == c)) q = 0.0;
      out[3 * i + k] = (uint8_t)(q * 255.0);
    }
  }
  unsigned int bits = __float_as_uint(fmaxf(w, 0.f));  // non-negative floats order like their bit patterns
  for (int off = 32; off > 0; off >>= 1) bits = max(bits, (unsigned int)__shfl_xor((int)bits, off, 64));
  if ((threadIdx.x & 63) == 0 && bits) atomicMax(max_w_bits, bits);
}

}  // namespace ptk

// ======================================================================== C ABI
using namespace ptk;

struct ptk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // scene
  bool have_scene = false;
  bool deep_tree = true;         // the tree has more than kShallowTreeDepth levels: the kernels' deep form runs
  bool tiny_scene = false;       // the tree and the primitive records fit the render kernel's LDS copy: its tiny form runs
  bool spill_stack = false;      // deep tree rendered with the shallow form's LDS stack plus overflow entries in private memory
  bool lds_tables = false;       // the material and light tables fit beside a kStackDepthTables-entry stack: the render kernel's table forms run
  bool tables_spill = false;     // ... and the tree has more than kTablesTreeDepth levels: with overflow entries
  bool have_scene_host = false;  // `host` still holds everything another context would need (ctx_upload_like)
  int32_t builder_asked = 0;
  HostScene host;  // kept for ptk_scene_info_get; texel pool is released after upload
  SceneView view;
  void* d_nodes = nullptr; void* d_prims = nullptr; void* d_shade = nullptr; void* d_materials = nullptr;
  void* d_lights = nullptr; void* d_textures = nullptr; void* d_texels = nullptr;
  uint64_t texel_bytes = 0;
  int32_t n_nodes = 0;
  int32_t builder = 0;    // what built the tree in use (PTK_BUILDER_*)
  double upload_ms = 0.0; // host wall time of the whole ptk_upload_scene call
  double tree_ms = 0.0;   // device time of the tree build, 0 for a host-built tree
  // camera / accumulation
  bool have_camera = false;
  CameraView cam;
  FlagView flags = {0, 1, 1, 1, 1, 1};
  f4* d_accum_own = nullptr;
  f4* d_accum = nullptr;  // own or bound
  size_t accum_pixels = 0;
  uint8_t* d_display = nullptr;  // W*H*3 display bytes staged for ptk_read_display_u8
  size_t display_capacity = 0;   // in pixels
  f4* d_partial = nullptr;  // per-chunk partial sums of the frame being rendered
  size_t partial_capacity = 0;  // in f4
  f4* d_static_scratch = nullptr;  // this device's copy of `scratch` (ptk_render_static)
  DeviceConsts* d_consts = nullptr;  // camera, switches and camera shift as the render kernel reads them (sync_consts)
  // staging of ptk_intersect_batch's host arrays; kept between calls, so that a batch no larger than an earlier
  // one (a single ray of Scene::find_intersection, say) allocates nothing
  float* d_stage_rays = nullptr;
  ptk_hit* d_stage_hits = nullptr;
  size_t stage_capacity = 0;  // in rays
  // counters and timing
  unsigned long long* d_counters = nullptr;
  unsigned int* d_unit_queue = nullptr;  // the render kernel's work-unit counters, one cache line per XCD
  int intersect_resident[2] = {0, 0};  // waves of the ray-batch kernel that fit the GPU at once: shallow form, deep form
  int render_resident[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ... and of the render kernel, per form (resident_workgroups)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, pool;
  double ms_last = 0.0, ms_total = 0.0;
  uint64_t launches = 0;
  uint64_t camera_rays = 0;  // of the render launches since ptk_reset_stats: rows x width x samples, counted here
  uint64_t batch_queries = 0;  // rays of the ray-batch launches since ptk_reset_stats that did not collect statistics
};

static std::string g_create_error;

#define PTK_HIP(call)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return PTK_ERR_HIP;                                                                        \
    }                                                                                            \
  } while (0)

static int fail(ptk_ctx* ctx, int code, const std::string& msg)
{
  if (ctx) ctx->err = msg; else g_create_error = msg;
  return code;
}

static void free_scene(ptk_ctx* ctx)
{
  void** ptrs[] = {&ctx->d_nodes, &ctx->d_prims, &ctx->d_shade, &ctx->d_materials, &ctx->d_lights, &ctx->d_textures, &ctx->d_texels};
  for (void** p : ptrs) { if (*p) (void)hipFree(*p); *p = nullptr; }
  ctx->have_scene = false;
}

static int resolve_timing(ptk_ctx* ctx)
{
  for (auto& pr : ctx->pending) {
    PTK_HIP(hipEventSynchronize(pr.second));
    float ms = 0.f;
    PTK_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
    ctx->ms_last = ms;
    ctx->ms_total += ms;
    ctx->pool.push_back(pr);
  }
  ctx->pending.clear();
  return PTK_OK;
}

static int begin_timed(ptk_ctx* ctx, std::pair<hipEvent_t, hipEvent_t>& ev)
{
  if (!ctx->pool.empty()) { ev = ctx->pool.back(); ctx->pool.pop_back(); }
  else {
    PTK_HIP(hipEventCreate(&ev.first));
    PTK_HIP(hipEventCreate(&ev.second));
  }
  PTK_HIP(hipEventRecord(ev.first, ctx->stream));
  return PTK_OK;
}

static int end_timed(ptk_ctx* ctx, std::pair<hipEvent_t, hipEvent_t>& ev)
{
  PTK_HIP(hipEventRecord(ev.second, ctx->stream));
  ctx->pending.push_back(ev);
  ctx->launches++;
  if (ctx->pending.size() > 4096) return resolve_timing(ctx);
  return PTK_OK;
}

template <class T>
static int upload(ptk_ctx* ctx, void** dst, const std::vector<T>& src)
{
  size_t bytes = std::max<size_t>(src.size() * sizeof(T), 16);
  PTK_HIP(hipMalloc(dst, bytes));
  if (!src.empty()) PTK_HIP(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return PTK_OK;
}

// Workgroups per tile.  A frame whose tiles alone give the chip only a few rounds of workgroups
// ends with most CUs idle while the slowest tiles finish; splitting every tile's samples over
// several workgroups shortens that tail.  Each chunk keeps at least kMinChunkSamples samples so
// that a lane still has work to refill with when its neighbours' paths end.
static int pick_chunks(int n_tiles, int samples, int64_t resident_waves)
{
  const int64_t kResidentWaves = resident_waves;
  constexpr int64_t kRounds = PTK_CHUNK_ROUNDS;
  constexpr int kMinChunkSamples = PTK_CHUNK_MIN_SAMPLES;
  int64_t waves = (int64_t)n_tiles;
  int64_t want = (kRounds * kResidentWaves + waves - 1) / waves;
  int64_t cap = std::max(1, samples / kMinChunkSamples);
  int64_t c = std::min<int64_t>(std::min(want, cap), kMaxChunks);
  if (const char* e = getenv("PTK_CHUNKS")) { int v = atoi(e); if (v >= 1) c = std::min<int64_t>(v, std::min<int64_t>(samples, 64)); }
  return (int)std::max<int64_t>(c, 1);
}

// Brings the device copy of the camera, the switches and the camera shift up to date (stream-ordered: launches
// already queued keep the values they were queued with).
static int sync_consts(ptk_ctx* ctx)
{
  DeviceConsts c;
  std::memset(&c, 0, sizeof c);
  c.cam = ctx->cam;
  c.fl = ctx->flags;
  if (ctx->have_scene && ctx->have_camera) {
    SceneView bounds;  // the bounds' grid is all camera_entry_shift looks at (the view's pointers are device addresses)
    std::memset(&bounds, 0, sizeof bounds);
    bounds.grid = ctx->view.grid;
    c.camera_shift = camera_entry_shift(bounds, ctx->cam);
  }
  PTK_HIP(hipMemcpyAsync(ctx->d_consts, &c, sizeof c, hipMemcpyHostToDevice, ctx->stream));
  return PTK_OK;
}

static void release_host_copy(ptk_ctx* ctx)
{
  std::vector<uint8_t>().swap(ctx->host.texels);
  std::vector<float>().swap(ctx->host.prim_boxes);
  if (ctx->builder == PTK_BUILDER_DEVICE_LBVH) std::vector<PrimRec>().swap(ctx->host.prims);  // id order: not the device's order
  ctx->have_scene_host = false;
}

using RenderKernel = void (*)(SceneView, const DeviceConsts*, RenderArgs, f4*, f4*, unsigned long long*, unsigned int*);
static RenderKernel render_kernel_for(int form, bool stats)
{
  switch (form) {
    case kFormSpillTables: return stats ? render_kernel<true, kFormSpillTables> : render_kernel<false, kFormSpillTables>;
    case kFormShallowTables: return stats ? render_kernel<true, kFormShallowTables> : render_kernel<false, kFormShallowTables>;
    case kFormSpill: return stats ? render_kernel<true, kFormSpill> : render_kernel<false, kFormSpill>;
    case kFormDeep: return stats ? render_kernel<true, kFormDeep> : render_kernel<false, kFormDeep>;
    case kFormTiny: return stats ? render_kernel<true, kFormTiny> : render_kernel<false, kFormTiny>;
    default: return stats ? render_kernel<true, kFormShallow> : render_kernel<false, kFormShallow>;
  }
}
// the form of the render kernel that the scene in use runs (see the forms' description at the kernel)
static int render_form(const ptk_ctx* ctx)
{
  return ctx->lds_tables ? (ctx->tables_spill ? kFormSpillTables : kFormShallowTables)
         : ctx->spill_stack ? kFormSpill : ctx->deep_tree ? kFormDeep : ctx->tiny_scene ? kFormTiny : kFormShallow;
}

// One-wave workgroups of `kernel` that the device holds at once: what the occupancy query says, corrected for the
// 1280-byte granules LDS is handed out in (which the query does not know), times the device's compute units,
// rounded down to a multiple of 8 (the render kernel's persistent launch gives every XCD the same number).
// Asked once per context and kernel form, when a scene that runs the form is uploaded (ptk_render_static must not be the first to ask).
static int resident_workgroups(ptk_ctx* ctx, RenderKernel kernel, int& cached)
{
  if (cached > 0) return cached;
  int per_cu = 0, cus = 0, lds_per_cu = 0;
  hipFuncAttributes fa;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess ||
      hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) != hipSuccess ||
      hipDeviceGetAttribute(&lds_per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, ctx->device) != hipSuccess) {
    (void)hipGetLastError();
    return cached = 256 * 4 * PTK_MIN_WAVES_SHALLOW;  // the MI355X figure
  }
  constexpr int kLdsGranule = 1280;
  const int lds = ((int)fa.sharedSizeBytes + kLdsGranule - 1) / kLdsGranule * kLdsGranule;
  if (lds > 0 && lds_per_cu > 0) per_cu = std::min(per_cu, lds_per_cu / lds);
  const int n = std::max(1, per_cu) * std::max(1, cus);
  return cached = std::max(8, n / 8 * 8);
}

// Second half of an upload: ctx->host holds the flattened scene (build_host_scene); build the tree on the device
// if that was asked for, copy everything to the GPU.  keep_host: leave the host copy whole (texels, boxes) so that
// another context of a group can be given the same scene (ctx_upload_like).
static int finish_upload(ptk_ctx* ctx, int32_t builder, bool keep_host = false)
{
  const auto t0 = std::chrono::steady_clock::now();
  HostScene& h = ctx->host;
  const bool on_device = !h.prim_boxes.empty();  // small scenes get their tree on the host whatever was asked
  ctx->builder = on_device ? PTK_BUILDER_DEVICE_LBVH : PTK_BUILDER_HOST_SAH;
  ctx->builder_asked = builder;
  int rc;
  if (on_device) {
    void *d_boxes = nullptr, *d_by_id = nullptr;
    if ((rc = upload(ctx, &d_boxes, h.prim_boxes))) { if (d_boxes) (void)hipFree(d_boxes); return rc; }
    if ((rc = upload(ctx, &d_by_id, h.prims))) { (void)hipFree(d_boxes); if (d_by_id) (void)hipFree(d_by_id); return rc; }
    DeviceTree tree;
    std::string why = build_device_tree(ctx->stream, (const float*)d_boxes, (const PrimRec*)d_by_id, h.n_prims, h.n_triangles, h.centroid_lo,
                                        h.centroid_hi, h.max_diagonal, h.grid, tree);
    (void)hipFree(d_boxes);
    (void)hipFree(d_by_id);
    if (!why.empty()) return fail(ctx, PTK_ERR_HIP, "ptk_upload_scene: " + why);
    ctx->d_nodes = tree.nodes;
    ctx->d_prims = tree.prims;
    h.max_depth = tree.max_depth;
    h.max_leaf = tree.max_leaf;
    ctx->n_nodes = tree.n_nodes;
    ctx->tree_ms = tree.build_ms;
  } else {
    ctx->n_nodes = (int32_t)h.nodes.size();
    ctx->tree_ms = 0.0;
  }
  // the kernels' per-lane stacks hold kStackDepth entries: a sentinel, one per level, one of scratch
  if (h.max_depth > kMaxTreeDepth || kMaxTreeDepth + 2 > kStackDepth) {
    free_scene(ctx);
    return fail(ctx, PTK_ERR_LIMIT, "ptk_upload_scene: the tree is deeper than the traversal stack");
  }
  ctx->deep_tree = h.max_depth > kShallowTreeDepth || getenv("PTK_FORCE_DEEP") != nullptr;
  ctx->spill_stack = ctx->deep_tree && getenv("PTK_DEEP_LDS") == nullptr;  // PTK_DEEP_LDS=1: the 27-entry LDS stack at 5 waves per SIMD
  ctx->tiny_scene = !ctx->deep_tree && h.max_depth <= kTinyTreeDepth && ctx->n_nodes <= kTinyNodes && h.n_prims <= kTinyPrims && h.n_prims > 0 &&
                    (int)h.materials.size() <= kTinyMaterials && (int)h.lights.size() <= kTinyLights &&
                    getenv("PTK_NO_TINY") == nullptr;
  ctx->lds_tables = !ctx->tiny_scene && (ctx->spill_stack || !ctx->deep_tree) &&
                    h.materials.size() * sizeof(MaterialRec) + h.lights.size() * sizeof(LightRec) <= (size_t)kTableBytes &&
                    getenv("PTK_NO_TABLES") == nullptr;
  ctx->tables_spill = ctx->lds_tables && (h.max_depth > kTablesTreeDepth || getenv("PTK_FORCE_DEEP") != nullptr);
  if (!on_device) {
    if ((rc = upload(ctx, &ctx->d_nodes, h.nodes))) return rc;
    if ((rc = upload(ctx, &ctx->d_prims, h.prims))) return rc;
  }
  if ((rc = upload(ctx, &ctx->d_shade, h.shade))) return rc;
  if ((rc = upload(ctx, &ctx->d_materials, h.materials))) return rc;
  if ((rc = upload(ctx, &ctx->d_lights, h.lights))) return rc;
  if ((rc = upload(ctx, &ctx->d_textures, h.textures))) return rc;
  if ((rc = upload(ctx, &ctx->d_texels, h.texels))) return rc;
  PTK_HIP(hipStreamSynchronize(ctx->stream));
  SceneView v = h.host_view();
  v.n_nodes = ctx->n_nodes;
  v.nodes = (const NodePair*)ctx->d_nodes;
  v.prims = (const PrimRec*)ctx->d_prims;
  v.shade = (const ShadeRec*)ctx->d_shade;
  v.materials = (const MaterialRec*)ctx->d_materials;
  v.lights = (const LightRec*)ctx->d_lights;
  v.textures = (const TextureRec*)ctx->d_textures + kGammaLutRecords;  // behind the gamma table
  v.texels = (const uint8_t*)ctx->d_texels;
  ctx->view = v;
  ctx->texel_bytes = h.texels.size();
  ctx->have_scene_host = keep_host;
  if (!keep_host) release_host_copy(ctx);  // the largest host copies are no longer needed
  ctx->upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->have_scene = true;
  (void)resident_workgroups(ctx, render_kernel_for(render_form(ctx), false), ctx->render_resident[render_form(ctx)]);
  return sync_consts(ctx);
}

static int upload_scene_impl(ptk_ctx* ctx, const ptk_scene_desc* scene, int32_t builder, bool keep_host)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!scene) return fail(ctx, PTK_ERR_INVALID, "ptk_upload_scene: scene is NULL");
  if (builder != PTK_BUILDER_HOST_SAH && builder != PTK_BUILDER_DEVICE_LBVH)
    return fail(ctx, PTK_ERR_INVALID, "ptk_upload_scene: unknown builder");
  PTK_HIP(hipSetDevice(ctx->device));
  PTK_HIP(hipStreamSynchronize(ctx->stream));
  free_scene(ctx);
  const auto t0 = std::chrono::steady_clock::now();
  std::string why = build_host_scene(*scene, ctx->host, builder == PTK_BUILDER_HOST_SAH);
  if (!why.empty()) return fail(ctx, PTK_ERR_INVALID, "ptk_upload_scene: " + why);
  ctx->upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return finish_upload(ctx, builder, keep_host);
}


extern "C" {

int ptk_create(int device, ptk_ctx** out)
{
  if (!out) return fail(nullptr, PTK_ERR_INVALID, "ptk_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, PTK_ERR_HIP, std::string("ptk_create: no HIP device available (") + hipGetErrorString(e) + ")");
  if (device < 0 || device >= n) return fail(nullptr, PTK_ERR_INVALID, "ptk_create: device index out of range");
  ptk_ctx* ctx = new (std::nothrow) ptk_ctx();
  if (!ctx) return fail(nullptr, PTK_ERR_INVALID, "ptk_create: out of host memory");
  ctx->device = device;
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counters, kCounterSlots * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemsetAsync(ctx->d_counters, 0, kCounterSlots * sizeof(unsigned long long), ctx->stream);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_unit_queue, kUnitQueueBytes);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_consts, sizeof(DeviceConsts));
  // with `device` current: the address of this device's copy of the static scratch array
  if (e == hipSuccess) e = hipGetSymbolAddress((void**)&ctx->d_static_scratch, HIP_SYMBOL(scratch));
  if (e != hipSuccess) {
    g_create_error = std::string("ptk_create: ") + hipGetErrorString(e);
    delete ctx;
    return PTK_ERR_HIP;
  }
  *out = ctx;
  return PTK_OK;
}

int ptk_destroy(ptk_ctx* ctx)
{
  if (!ctx) return PTK_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_scene(ctx);
  if (ctx->d_accum_own) (void)hipFree(ctx->d_accum_own);
  if (ctx->d_partial) (void)hipFree(ctx->d_partial);
  if (ctx->d_display) (void)hipFree(ctx->d_display);
  if (ctx->d_stage_rays) (void)hipFree(ctx->d_stage_rays);
  if (ctx->d_stage_hits) (void)hipFree(ctx->d_stage_hits);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  if (ctx->d_unit_queue) (void)hipFree(ctx->d_unit_queue);
  if (ctx->d_consts) (void)hipFree(ctx->d_consts);
  for (auto& pr : ctx->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto& pr : ctx->pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return PTK_OK;
}

const char* ptk_last_error(const ptk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int ptk_upload_scene_ex(ptk_ctx* ctx, const ptk_scene_desc* scene, int32_t builder)
{
  return upload_scene_impl(ctx, scene, builder, false);
}

int ptk_upload_scene(ptk_ctx* ctx, const ptk_scene_desc* scene)
{
  int32_t builder = PTK_BUILDER_HOST_SAH;
  if (const char* e = getenv("PTK_BUILDER")) {
    if (std::strcmp(e, "device") == 0) builder = PTK_BUILDER_DEVICE_LBVH;
    else if (std::strcmp(e, "host") != 0 && ctx) return fail(ctx, PTK_ERR_INVALID, "PTK_BUILDER must be host or device");
  }
  return ptk_upload_scene_ex(ctx, scene, builder);
}

int ptk_scene_info_get(const ptk_ctx* ctx, ptk_scene_info* out)
{
  if (!ctx || !out || !ctx->have_scene) return PTK_ERR_INVALID;
  const HostScene& h = ctx->host;
  out->n_primitives = h.n_prims;
  out->n_lights = (int32_t)h.lights.size();
  out->n_nodes = ctx->n_nodes;
  out->max_depth = h.max_depth;
  out->max_leaf_size = h.max_leaf;
  out->traversal_kind = render_form(ctx);
  out->node_bytes = (uint64_t)ctx->n_nodes * sizeof(NodePair);
  out->prim_bytes = (uint64_t)std::max(h.n_prims, 1) * sizeof(PrimRec);
  out->shading_bytes = h.shade.size() * sizeof(ShadeRec) + h.materials.size() * sizeof(MaterialRec) + h.lights.size() * sizeof(LightRec);
  out->texture_bytes = ctx->texel_bytes;
  out->builder = ctx->builder;
  out->tree_from_cache = ctx->host.tree_from_cache ? 1 : 0;
  out->upload_ms = ctx->upload_ms;
  out->device_build_ms = ctx->tree_ms;
  return PTK_OK;
}

int ptk_set_camera(ptk_ctx* ctx, const ptk_camera* camera)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!camera) return fail(ctx, PTK_ERR_INVALID, "ptk_set_camera: camera is NULL");
  if (camera->width <= 0 || camera->height <= 0 || (int64_t)camera->width * camera->height > (int64_t)1 << 28)
    return fail(ctx, PTK_ERR_LIMIT, "ptk_set_camera: image size must be 1 .. 2^28 pixels");
  PTK_HIP(hipSetDevice(ctx->device));
  PTK_HIP(hipStreamSynchronize(ctx->stream));
  size_t pixels = (size_t)camera->width * (size_t)camera->height;
  if (pixels != ctx->accum_pixels || !ctx->d_accum_own) {
    if (ctx->d_accum_own) (void)hipFree(ctx->d_accum_own);
    // nothing may point at the freed buffer if the allocation below fails
    ctx->d_accum_own = nullptr;
    ctx->d_accum = nullptr;
    ctx->accum_pixels = 0;
    ctx->have_camera = false;
    PTK_HIP(hipMalloc((void**)&ctx->d_accum_own, pixels * sizeof(f4)));
    ctx->accum_pixels = pixels;
    ctx->d_accum = ctx->d_accum_own;  // a caller's buffer bound for another image size is no longer valid
  }
  // the buffer in use is cleared, whether it is the context's own or one the caller bound (same size)
  PTK_HIP(hipMemsetAsync(ctx->d_accum, 0, pixels * sizeof(f4), ctx->stream));
  ctx->cam = make_camera_view(*camera);
  ctx->have_camera = true;
  return sync_consts(ctx);
}

int ptk_set_flags(ptk_ctx* ctx, const ptk_flags* flags)
{
  if (!ctx) return PTK_ERR_INVALID;
  ctx->flags = make_flag_view(flags);
  PTK_HIP(hipSetDevice(ctx->device));
  return sync_consts(ctx);
}

int ptk_bind_accum(ptk_ctx* ctx, float* device_rgbw)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!ctx->have_camera) return fail(ctx, PTK_ERR_INVALID, "ptk_bind_accum: set the camera first");
  if (device_rgbw && ((uintptr_t)device_rgbw & 15)) return fail(ctx, PTK_ERR_INVALID, "ptk_bind_accum: buffer must be 16-byte aligned");
  ctx->d_accum = device_rgbw ? (f4*)device_rgbw : ctx->d_accum_own;
  return PTK_OK;
}

int ptk_reset_accum(ptk_ctx* ctx)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!ctx->have_camera) return fail(ctx, PTK_ERR_INVALID, "ptk_reset_accum: set the camera first");
  PTK_HIP(hipSetDevice(ctx->device));
  PTK_HIP(hipMemsetAsync(ctx->d_accum, 0, ctx->accum_pixels * sizeof(f4), ctx->stream));
  return PTK_OK;
}

// Body of ptk_render and ptk_render_static.  static_scratch: take the partial-sum buffer from the device's
// static array, record no timing events and call nothing that allocates.
static int render_common(ptk_ctx* ctx, const ptk_render_params* p, bool static_scratch)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!p) return fail(ctx, PTK_ERR_INVALID, "ptk_render: params is NULL");
  if (!ctx->have_scene || !ctx->have_camera) return fail(ctx, PTK_ERR_INVALID, "ptk_render: upload a scene and set a camera first");
  if (p->samples < 0 || p->max_bounce < 0 || p->sample_stride < 1 || p->sample_first < 0)
    return fail(ctx, PTK_ERR_INVALID, "ptk_render: samples, max_bounce, sample_first must be >= 0 and sample_stride >= 1");
  if ((int64_t)p->sample_first + (int64_t)p->samples * p->sample_stride > (int64_t)INT32_MAX)
    return fail(ctx, PTK_ERR_LIMIT, "ptk_render: sample index exceeds 2^31 - 1");
  if (p->samples > (1 << 25)) return fail(ctx, PTK_ERR_LIMIT, "ptk_render: at most 2^25 samples per call");
  // per-lane counts are 32-bit registers (summed over a wave in 64 bits): a lane's share of a launch that also
  // counts box and primitive tests must stay below 2^32 of them
  if (p->collect_stats && p->samples > (1 << 20)) return fail(ctx, PTK_ERR_LIMIT, "ptk_render: at most 2^20 samples per call with collect_stats");
  RenderArgs ra;
  ra.samples = p->samples; ra.seed = p->seed;
  ra.max_bounce = p->max_bounce < (1 << 23) ? p->max_bounce : (1 << 23);  // the range of the path state's depth field
  ra.sample_first = p->sample_first; ra.sample_stride = p->sample_stride;
  ra.row_begin = p->row_begin;
  ra.row_end = p->row_end <= 0 ? ctx->cam.height : p->row_end;
  if (ra.row_begin < 0 || ra.row_end > ctx->cam.height || ra.row_begin > ra.row_end)
    return fail(ctx, PTK_ERR_INVALID, "ptk_render: row range outside the image");
  ra.band_rows = 0; ra.band_first = 0; ra.band_stride = 1;
  int launch_rows = ra.row_end - ra.row_begin;
  if (p->band_rows != 0) {
    if (p->band_rows < 0 || p->band_rows % kTile != 0 || p->band_first < 0 || p->band_stride < 1)
      return fail(ctx, PTK_ERR_INVALID, "ptk_render: band_rows must be a positive multiple of 8, band_first >= 0, band_stride >= 1");
    if (ra.row_begin != 0 || ra.row_end != ctx->cam.height)
      return fail(ctx, PTK_ERR_INVALID, "ptk_render: interleaved bands cover the whole image: leave row_begin and row_end 0");
    const int n_bands = (ctx->cam.height + p->band_rows - 1) / p->band_rows;
    const int mine = p->band_first < n_bands ? (n_bands - p->band_first + p->band_stride - 1) / p->band_stride : 0;
    ra.band_rows = p->band_rows; ra.band_first = p->band_first; ra.band_stride = p->band_stride;
    launch_rows = mine * p->band_rows;
  }
  if (p->samples == 0 || launch_rows == 0) return PTK_OK;
  ra.tiles_x = (ctx->cam.width + kTile - 1) / kTile;
  int tiles_y = (launch_rows + kTile - 1) / kTile;
  ra.rows = tiles_y * kTile;
  ra.n_tiles = ra.tiles_x * tiles_y;
  PTK_HIP(hipSetDevice(ctx->device));
  const int form = render_form(ctx);
  RenderKernel kernel = render_kernel_for(form, p->collect_stats != 0);
  const int64_t resident = resident_workgroups(ctx, kernel, ctx->render_resident[form]);  // one wave each
  ra.chunks = pick_chunks(ra.n_tiles, p->samples, resident);
  size_t band_pixels = (size_t)ra.rows * (size_t)ctx->cam.width;
  f4* partial = ctx->d_partial;
  if (static_scratch) {
    partial = ctx->d_static_scratch;
    const size_t need = ra.chunks > 1 ? band_pixels * (size_t)ra.chunks : 0;
    if (need * sizeof(f4) > kStaticScratchBytes)
      return fail(ctx, PTK_ERR_LIMIT, "ptk_render_static: " + std::to_string(band_pixels) + " pixels in " + std::to_string(ra.chunks) +
                                          " sample chunks need " + std::to_string(need * sizeof(f4)) + " bytes of partial sums, the static buffer holds " +
                                          std::to_string(kStaticScratchBytes) + ": render fewer rows (row_begin / row_end) or fewer samples per call");
  } else if (ra.chunks > 1) {
    size_t need = band_pixels * (size_t)ra.chunks;
    if (need > ctx->partial_capacity) {
      PTK_HIP(hipStreamSynchronize(ctx->stream));  // an earlier frame may still be folding the old buffer
      if (ctx->d_partial) (void)hipFree(ctx->d_partial);
      ctx->d_partial = nullptr;
      ctx->partial_capacity = 0;
      PTK_HIP(hipMalloc((void**)&ctx->d_partial, need * sizeof(f4)));
      ctx->partial_capacity = need;
    }
    partial = ctx->d_partial;
  }
  {  // camera rays of this launch: every pixel of every row it covers, `samples` times (none when nothing may bounce: still counted, as Camera::get_ray is still called)
    uint64_t rows = 0;
    for (int r = 0; r < launch_rows; r++) rows += launch_row_to_image_row(ra, r) < ra.row_end ? 1u : 0u;
    ctx->camera_rays += rows * (uint64_t)ctx->cam.width * (uint64_t)p->samples;
  }
  std::pair<hipEvent_t, hipEvent_t> ev;
  int rc = static_scratch ? PTK_OK : begin_timed(ctx, ev);  // timing events may have to be created
  if (rc) return rc;
  // More units than fit the GPU at once: as many workgroups as fit, which draw the units from per-XCD counters.
  const int64_t n_units = (int64_t)ra.n_tiles * ra.chunks;
  // (+0.2-0.4 % on the mesh scenes; the Cornell box's whole-scene-in-LDS form loses as much and keeps one workgroup per unit)
  ra.persistent = n_units > resident && !ctx->tiny_scene && getenv("PTK_NO_PERSISTENT") == nullptr ? 1 : 0;
  if (ra.persistent) PTK_HIP(hipMemsetAsync(ctx->d_unit_queue, 0, kUnitQueueBytes, ctx->stream));
  dim3 grid((unsigned)(ra.persistent ? resident : n_units)), block(kBlock);
  hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, ctx->view, ctx->d_consts, ra, ctx->d_accum, partial, ctx->d_counters,
                     ctx->d_unit_queue);
  PTK_HIP(hipGetLastError());
  if (static_scratch) ctx->launches++;
  else rc = end_timed(ctx, ev);
  if (rc) return rc;
  if (ra.chunks > 1) {
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)((band_pixels + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_accum,
                       partial, ra, ctx->cam.width);
    PTK_HIP(hipGetLastError());
  }
  return PTK_OK;
}

int ptk_render(ptk_ctx* ctx, const ptk_render_params* p) { return render_common(ctx, p, false); }

int ptk_render_static(ptk_ctx* ctx, const ptk_render_params* p) { return render_common(ctx, p, true); }

int ptk_synchronize(ptk_ctx* ctx)
{
  if (!ctx) return PTK_ERR_INVALID;
  PTK_HIP(hipSetDevice(ctx->device));
  PTK_HIP(hipStreamSynchronize(ctx->stream));
  return resolve_timing(ctx);
}

int ptk_read_accum(ptk_ctx* ctx, float* rgb, int32_t* total_samples)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!rgb) return fail(ctx, PTK_ERR_INVALID, "ptk_read_accum: rgb is NULL");
  if (!ctx->have_camera) return fail(ctx, PTK_ERR_INVALID, "ptk_read_accum: set the camera first");
  PTK_HIP(hipSetDevice(ctx->device));
  std::vector<f4> tmp(ctx->accum_pixels);
  PTK_HIP(hipMemcpyAsync(tmp.data(), ctx->d_accum, tmp.size() * sizeof(f4), hipMemcpyDeviceToHost, ctx->stream));
  int rc = ptk_synchronize(ctx);
  if (rc) return rc;
  float max_w = 0.f;
  for (size_t i = 0; i < tmp.size(); i++) {
    float w = tmp[i].w;
    float inv = w > 0.f ? 1.0f / w : 0.f;
    rgb[3 * i + 0] = tmp[i].x * inv; rgb[3 * i + 1] = tmp[i].y * inv; rgb[3 * i + 2] = tmp[i].z * inv;
    if (w > max_w) max_w = w;
  }
  if (total_samples) *total_samples = (int32_t)max_w;
  return PTK_OK;
}

int ptk_read_display_u8(ptk_ctx* ctx, uint8_t* rgb8, int32_t* total_samples)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!rgb8) return fail(ctx, PTK_ERR_INVALID, "ptk_read_display_u8: rgb8 is NULL");
  if (!ctx->have_camera) return fail(ctx, PTK_ERR_INVALID, "ptk_read_display_u8: set the camera first");
  PTK_HIP(hipSetDevice(ctx->device));
  const size_t n = ctx->accum_pixels;
  if (ctx->display_capacity < n) {
    PTK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_display) (void)hipFree(ctx->d_display);
    ctx->d_display = nullptr;
    ctx->display_capacity = 0;
    PTK_HIP(hipMalloc((void**)&ctx->d_display, n * 3 + 8));  // + the largest sample count, 4-byte aligned
    ctx->display_capacity = n;
  }
  float* d_max_w = (float*)(ctx->d_display + ((n * 3 + 3) & ~(size_t)3));
  PTK_HIP(hipMemsetAsync(d_max_w, 0, sizeof(float), ctx->stream));
  hipLaunchKernelGGL(display_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_accum, ctx->d_display, n,
                     (unsigned int*)d_max_w);
  PTK_HIP(hipGetLastError());
  float max_w = 0.f;
  PTK_HIP(hipMemcpyAsync(rgb8, ctx->d_display, n * 3, hipMemcpyDeviceToHost, ctx->stream));
  PTK_HIP(hipMemcpyAsync(&max_w, d_max_w, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  int rc = ptk_synchronize(ctx);
  if (rc) return rc;
  if (total_samples) *total_samples = (int32_t)max_w;
  return PTK_OK;
}

static int launch_intersect(ptk_ctx* ctx, const float* d_rays, int32_t n, ptk_hit* d_hits, bool stats)
{
  std::pair<hipEvent_t, hipEvent_t> ev;
  int rc = begin_timed(ctx, ev);
  if (rc) return rc;
  auto kernel = ctx->deep_tree ? (stats ? intersect_kernel<true, true> : intersect_kernel<false, true>)
                               : (stats ? intersect_kernel<true, false> : intersect_kernel<false, false>);
  // One round of resident waves, fewer for a batch that does not fill them (PTK_INTERSECT_GRID overrides).
  int& resident = ctx->intersect_resident[ctx->deep_tree ? 1 : 0];
  if (resident <= 0) {
    int per_cu = 0, cus = 0, lds_per_cu = 0;
    PTK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0));
    PTK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    // LDS is handed out in granules of 1280 bytes here, which the occupancy query does not know (it answers 27 where
    // 25 workgroups fit); the two extra workgroups per CU would run after the others have ended.
    hipFuncAttributes fa;
    PTK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)));
    PTK_HIP(hipDeviceGetAttribute(&lds_per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, ctx->device));
    constexpr int kLdsGranule = 1280;
    const int lds = ((int)fa.sharedSizeBytes + kLdsGranule - 1) / kLdsGranule * kLdsGranule;
    if (lds > 0 && lds_per_cu > 0) per_cu = std::min(per_cu, lds_per_cu / lds);
    resident = std::max(1, per_cu) * std::max(1, cus);
    if (const char* e = getenv("PTK_INTERSECT_GRID")) resident = std::max(1, atoi(e));
  }
  // Rays per draw: two draws per wave for a small batch (so that all waves have work), kIntersectRunMax for a large one.
  int64_t run = ((int64_t)n / (2 * (int64_t)resident) + kBlock - 1) / kBlock * kBlock;
  run = std::min<int64_t>(std::max<int64_t>(run, kBlock), kIntersectRunMax);
  if (const char* e = getenv("PTK_INTERSECT_RUN")) run = std::max(1, atoi(e));
  dim3 grid((unsigned)std::min<int64_t>(((int64_t)n + run - 1) / run, resident)), block(kBlock);
  PTK_HIP(hipMemsetAsync(ctx->d_counters + kIntersectQueueSlot, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, ctx->view, d_rays, n, d_hits, ctx->d_counters, (int)run);
  PTK_HIP(hipGetLastError());
  if (!stats) ctx->batch_queries += (uint64_t)n;
  return end_timed(ctx, ev);
}

int ptk_intersect_batch_device(ptk_ctx* ctx, const float* device_rays, int32_t n, ptk_hit* device_hits)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!ctx->have_scene) return fail(ctx, PTK_ERR_INVALID, "ptk_intersect_batch_device: upload a scene first");
  if (n < 0 || (n > 0 && (!device_rays || !device_hits))) return fail(ctx, PTK_ERR_INVALID, "ptk_intersect_batch_device: bad arguments");
  if (((uintptr_t)device_rays & 7) || ((uintptr_t)device_hits & 3))
    return fail(ctx, PTK_ERR_INVALID, "ptk_intersect_batch_device: rays must be 8-byte aligned, hits 4-byte aligned");
  if (n == 0) return PTK_OK;
  PTK_HIP(hipSetDevice(ctx->device));
  return launch_intersect(ctx, device_rays, n, device_hits, false);
}

int ptk_intersect_batch(ptk_ctx* ctx, const ptk_ray* rays, int32_t n, ptk_hit* hits)
{
  if (!ctx) return PTK_ERR_INVALID;
  if (!ctx->have_scene) return fail(ctx, PTK_ERR_INVALID, "ptk_intersect_batch: upload a scene first");
  if (n < 0 || (n > 0 && (!rays || !hits))) return fail(ctx, PTK_ERR_INVALID, "ptk_intersect_batch: bad arguments");
  if (n == 0) return PTK_OK;
  PTK_HIP(hipSetDevice(ctx->device));
  if ((size_t)n > ctx->stage_capacity) {
    PTK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage_rays) (void)hipFree(ctx->d_stage_rays);
    if (ctx->d_stage_hits) (void)hipFree(ctx->d_stage_hits);
    ctx->d_stage_rays = nullptr; ctx->d_stage_hits = nullptr; ctx->stage_capacity = 0;
    const size_t cap = std::max<size_t>((size_t)n, 64);
    PTK_HIP(hipMalloc((void**)&ctx->d_stage_rays, cap * 6 * sizeof(float)));
    PTK_HIP(hipMalloc((void**)&ctx->d_stage_hits, cap * sizeof(ptk_hit)));
    ctx->stage_capacity = cap;
  }
  float one[6];
  std::vector<float> many;
  float* fr = one;
  if (n > 1) { many.resize(6 * (size_t)n); fr = many.data(); }
  for (size_t i = 0; i < (size_t)n; i++)
    for (int k = 0; k < 3; k++) { fr[6 * i + k] = (float)rays[i].origin[k]; fr[6 * i + 3 + k] = (float)rays[i].direction[k]; }
  PTK_HIP(hipMemcpyAsync(ctx->d_stage_rays, fr, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  int rc = launch_intersect(ctx, ctx->d_stage_rays, n, ctx->d_stage_hits, true);
  if (rc) return rc;
  PTK_HIP(hipMemcpyAsync(hits, ctx->d_stage_hits, (size_t)n * sizeof(ptk_hit), hipMemcpyDeviceToHost, ctx->stream));
  PTK_HIP(hipStreamSynchronize(ctx->stream));  // also covers `fr`: the copy in has completed
  return resolve_timing(ctx);
}

int ptk_get_stats(ptk_ctx* ctx, ptk_stats* out)
{
  if (!ctx || !out) return PTK_ERR_INVALID;
  int rc = ptk_synchronize(ctx);
  if (rc) return rc;
  unsigned long long c[kCounterSlots];
  PTK_HIP(hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
  out->camera_rays = ctx->camera_rays; out->segments = c[1]; out->queries = c[2] + ctx->batch_queries; out->box_tests = c[3]; out->prim_tests = c[4];
  PTK_PHASE(if (getenv("PTK_PHASE_STATS")) {
    auto pct = [](unsigned long long lanes, unsigned long long iters) { return iters ? 100.0 * (double)lanes / (64.0 * (double)iters) : 0.0; };
    fprintf(stderr, "[ptk phases] descend: %llu wave-steps, %.1f%% lanes | leaves: %llu wave-steps, %.1f%% lanes at a leaf, %.1f%% at an inner node | advance: %llu wave-steps, %.1f%% lanes | leaf-loop passes: %llu (%.2f per leaf phase), %.1f%% lanes\n",
            c[8], pct(c[9], c[8]), c[10], pct(c[11], c[10]), pct(c[12], c[10]), c[13], pct(c[14], c[13]), c[15], c[10] ? (double)c[15] / c[10] : 0.0,
            pct(c[4], c[15]));
    const double tsum = (double)(c[16] + c[17] + c[18] + c[19] + c[20]);
    if (tsum > 0)
      fprintf(stderr, "[ptk phase clocks] descend %.1f%% (%.0f per wave-step) | leaves %.1f%% (%.0f) | lit %.1f%% | shade %.1f%% (%.0f per shade step) | refill %.1f%%\n",
              100.0 * c[16] / tsum, c[8] ? (double)c[16] / c[8] : 0.0, 100.0 * c[17] / tsum, c[10] ? (double)c[17] / c[10] : 0.0,
              100.0 * c[18] / tsum, 100.0 * c[19] / tsum, c[13] ? (double)c[19] / c[13] : 0.0, 100.0 * c[20] / tsum);
  })
  out->kernel_ms_last = ctx->ms_last; out->kernel_ms_total = ctx->ms_total; out->launches = ctx->launches;
  return PTK_OK;
}

int ptk_reset_stats(ptk_ctx* ctx)
{
  if (!ctx) return PTK_ERR_INVALID;
  int rc = ptk_synchronize(ctx);
  if (rc) return rc;
  PTK_HIP(hipMemsetAsync(ctx->d_counters, 0, kCounterSlots * sizeof(unsigned long long), ctx->stream));
  ctx->ms_last = ctx->ms_total = 0.0;
  ctx->launches = 0;
  ctx->camera_rays = 0;
  ctx->batch_queries = 0;
  return PTK_OK;
}

void* ptk_stream(ptk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

}  // extern "C"

// ---- hooks for ptk_group.hip (ptk_internal.h) -------------------------------------------------------------
namespace ptk {

int ctx_device(const ptk_ctx* ctx) { return ctx->device; }
float* ctx_accum(ptk_ctx* ctx) { return reinterpret_cast<float*>(ctx->d_accum); }
size_t ctx_accum_pixels(const ptk_ctx* ctx) { return ctx->have_camera ? ctx->accum_pixels : 0; }
void ctx_set_error(ptk_ctx* ctx, const std::string& msg) { ctx->err = msg; }
int ctx_upload_keep_host(ptk_ctx* ctx, const ptk_scene_desc* scene, int32_t builder) { return upload_scene_impl(ctx, scene, builder, true); }
void ctx_release_host_copy(ptk_ctx* ctx) { release_host_copy(ctx); }
void ctx_drop_scene(ptk_ctx* ctx)
{
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_scene(ctx);
}

// Gives `to` the scene that `from` holds on the host side (the flattened records and, for a host-built tree,
// the tree) and uploads it to `to`'s GPU: the tree is built once for all GPUs of a group.
int ctx_upload_like(ptk_ctx* to, ptk_ctx* from)
{
  ptk_ctx* ctx = to;
  if (!from->have_scene_host) return fail(ctx, PTK_ERR_INVALID, "ptk_group_upload_scene: the first context holds no scene");
  PTK_HIP(hipSetDevice(ctx->device));
  PTK_HIP(hipStreamSynchronize(ctx->stream));
  free_scene(ctx);
  ctx->host = from->host;
  return finish_upload(ctx, from->builder_asked);
}

}  // namespace ptk
