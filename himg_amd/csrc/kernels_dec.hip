// kernels_dec.hip -- HIMG decode path as hand-written HIP for gfx950 (MI355X).
//
// Pipeline (all on device, batched over frames):
//   k_dec_parse       RIFF chunk headers, mapping tables, QCFG, both Huffman trees,
//                     decode tables (decoder.cpp:144-290,428-461, huffman_dec.cpp:152-229)
//   k_dec_rowwalk     index of the FRES block rows (huffman_dec.cpp:232-248), side stream
//   k_lres_spec, k_lres_fix, k_lres_write, k_lres_finish
//                     the LRES stream, all chunks in parallel (huffman_dec.cpp:274-418)
//   k_lres_unpredict  inverse low-res prediction              (downsampled.cpp:318-382)
//   k_row_count       fixpoint rounds of every FRES block row (huffman_dec.cpp:274-418)
//   k_dec_row_fused   per block row: write pass into LDS, dequantise, inverse WHT,
//                     low-res add, clamp, colour inverse, pixel stores
//                     (decoder.cpp:331-426, quantize.cpp:153-165, hadamard.cpp:90-103,
//                     ycbcr.cpp:54-82)
//   k_dec_huff, k_tile_inv
//                     the same two steps through HBM for rows wider than the LDS, and
//                     the serial LRES fallback
// One entropy-decode engine serves all of them: see "Entropy decoding" below.
#include "himg_dev.h"
#include "loop_counts.h"

#include <cstdlib>
#include <type_traits>

namespace himg_dev {

__device__ static constexpr uint8_t kScanD[64] = {
    0,  1,  9,  8,  16, 17, 18, 10, 2,  3,  11, 19, 27, 26, 25, 24,
    32, 33, 34, 35, 36, 28, 20, 12, 4,  5,  13, 21, 29, 37, 45, 44,
    43, 42, 41, 40, 48, 49, 50, 51, 52, 53, 54, 46, 38, 30, 22, 14,
    6,  7,  15, 23, 31, 39, 47, 55, 63, 62, 61, 60, 59, 58, 57, 56};

// Inverse of kScanD: kInvScanD[row-major position] = index in the coefficient scan.
__device__ static constexpr uint8_t kInvScanD[64] = {
    0, 1, 8, 9, 24, 25, 48, 49, 3, 2, 7, 10, 23, 26, 47, 50, 4, 5, 6, 11, 22, 27, 46, 51, 15, 14, 13, 12, 21, 28, 45, 52, 16, 17, 18, 19, 20, 29, 44, 53, 35, 34, 33, 32, 31, 30, 43, 54, 36, 37, 38, 39, 40, 41, 42, 55, 63, 62, 61, 60, 59, 58, 57, 56};

// Device status values (host maps them to HIMG_ERR_*): 0 ok, 1 geometry
// mismatch, 3 outside the built scope, 4 the reference would return false.
// A format error also carries the decoder stage that failed in bits 4..7 and
// "the Huffman layer complained first" in bit 8, so that the C++ wrapper can
// print the same lines as the reference (decoder.cpp:96-135,232,287,345).
constexpr int kStGeom = 1, kStUnsupported = 3, kStFormat = 4;
__device__ __host__ constexpr int fmt_err(int stage, int huff) { return kStFormat | (stage << 4) | (huff << 8); }
constexpr int kMaxDepth = 32;   // deepest code the decoder walks (the reference encoder keeps codes in uint32_t)
constexpr int kMaxNodes = 2 * kNumSym - 1;
constexpr int kMaxSlow = 128;   // kLutBits-bit prefixes that get a second-level sub-table

__device__ __forceinline__ int clamp255d(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

__device__ __forceinline__ uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// Forward chunk search (decoder.cpp:428-461).  Returns false when not found.
__device__ bool find_chunk(const uint8_t *p, uint32_t n, uint32_t *idx, uint32_t tag,
                           uint32_t *size) {
  for (;;) {
    if (*idx + 8 > n) return false;
    const uint32_t t = rd32(p + *idx), sz = rd32(p + *idx + 4);
    *idx += 8;
    if (sz > 0x7fffffffu || (unsigned long long)*idx + sz > n) return false;
    if (t == tag) { *size = sz; return true; }
    *idx += sz;
  }
}

// Pre-order tree recovery (huffman_dec.cpp:152-213), in parallel.  The reference reads
// the serialised tree node by node (a branch is one 0 bit, a leaf a 1 bit and nine symbol
// bits); as a single-lane loop that is 250 instructions of divergent control per node
// and 120 us per frame.  Here:
//   A  where the nodes start.  One lane per 32-bit word runs the walk over its word
//      from each of the ten possible entry states (0..9 = bits of a leaf left over from
//      the previous word): exit state and node count per entry state.  One lane chains
//      the (at most 96) words; the word lanes then write their nodes' start bits.
//   B  the tree.  open[k] = stack depth of the reference's walk after node k (a prefix
//      sum of +1 / -1); the tree ends at the first node where it is 0.  A node that
//      follows a branch is its first child; one that follows a leaf is the second child
//      of the nearest earlier branch j with open[j] == open[k - 1] + 1 (the branch whose
//      pending child sat on top of the stack).  Depth and code (LSB = first choice)
//      follow from the parent links by pointer jumping.
// Per node the low bits of its code and its depth are kept in aux[] for the table
// fills; the nodes go to LDS in the packed form the tree walks read (child a |
// child b << 10 | (symbol + 1) << 20, 1023 = no child).
struct TreeAux { uint32_t code; int32_t depth; };

constexpr int kTreeWords = kTreeStride / 4;       // words of a staged serialised tree
constexpr int kNodeCap = kMaxNodes + 1;           // node slots (a tree of more nodes is rejected)
struct TreeScratch {                              // phases A / B of one tree
  unsigned long long outs[kTreeWords], cnts[kTreeWords];   // 4 / 6 bits per entry state
  uint32_t went[kTreeWords];                      // entry state | first node << 4
  uint16_t pos[kNodeCap + 6];                     // start bit | leaf << 15
  int16_t open[2][kNodeCap + 6];                  // prefix sums (two buffers)
  __attribute__((aligned(16))) int16_t key[kNodeCap + 14];   // open[] of branches, 0 for leaves
  uint16_t anc[kNodeCap + 6];
  uint16_t second[kNodeCap + 6];                  // second child of a branch
  int32_t total, first_bad, kend, kdeep;
};

// One entry of the LMAP / FMAP mapping table from bytes staged in LDS (mapper.cpp:127-157):
// entries 1..n1 are one byte each, the others two.
__device__ __forceinline__ bool map_ok(const uint8_t *in, uint32_t size) {
  return size >= 1 && in[0] <= 127 && (uint32_t)(1 + in[0] + 2 * (127 - in[0])) == size;
}
__device__ __forceinline__ int16_t map_entry(const uint8_t *in, int i) {
  const int n1 = in[0] <= 127 ? in[0] : 127;
  if (i == 0) return 0;
  if (i <= n1) return (int16_t)in[i];
  const uint8_t *q = in + 1 + n1 + 2 * (i - n1 - 1);
  return (int16_t)(uint16_t)(q[0] | (q[1] << 8));
}

// LUT entry: [8:0] symbol | [9] "continue at node" flag | [15:10] code bits |
// [19:16] RLE extra bits of the symbol | [31:20] node index (the leaf itself, or
// for flagged entries the branch reached after kLutBits bits).
__device__ __forceinline__ int rle_extra_bits(int sym) {
  return sym < 257 ? 0 : sym == 257 ? 2 : sym == 258 ? 4 : sym == 259 ? 8 : 14;
}
__device__ __forceinline__ uint32_t lut_leaf(int sym, int bits) {
  return (uint32_t)sym | ((uint32_t)bits << 10) | ((uint32_t)rle_extra_bits(sym) << 16);
}
__device__ __forceinline__ uint32_t lut_node(int node, int bits) {
  return 512u | ((uint32_t)bits << 10) | ((uint32_t)node << 20);
}
// Step word of the decoder (see "Entropy decoding" below): [4:0] code bits to
// consume, [9:5] extra bits that follow, [18:10] symbols produced before the
// extra-bits value is added, [22:19] code bits / [25:23] class of the group's first
// token, [31:27] = [4:0] + [9:5].
__device__ __forceinline__ uint32_t grp_y(uint32_t tb, uint32_t eb, uint32_t cb, uint32_t s_tb, uint32_t s_class) {
  return tb | (eb << 5) | (cb << 10) | (s_tb << 19) | (s_class << 23) | ((tb + eb) << 27);
}
// Second-level entry of a token whose code is longer than kLutBits (the first
// kLutBits bits are consumed before it is read): [4:0] the remaining code bits (>= 1,
// which tells it from a node reference), [9:5] extra bits, [18:10] run base / 1,
// [26:19] the literal byte.
__device__ __forceinline__ uint32_t sub_leaf(int sym, uint32_t rest_bits) {
  const uint32_t c = sym < 256 ? 0u : (uint32_t)sym - 255u;
  const uint32_t eb = (0xE84200u >> (4u * c)) & 15u;
  const unsigned long long kBases = 1ull | (2ull << 9) | (3ull << 18) | (7ull << 27) | (23ull << 36) | (279ull << 45);
  const uint32_t cb = (uint32_t)(kBases >> (9u * c)) & 511u;
  return rest_bits | (eb << 5) | (cb << 10) | ((sym < 256 ? (uint32_t)sym : 0u) << 19);
}

// ---------------------------------------------------------------------------
// k_dec_parse: one 1024-thread workgroup per frame.
//   1. thread 0 walks the chunk headers only (decoder.cpp:144-290): RIFF, FRMT,
//      LMAP, LRES, QCFG, FMAP, FRES -- a chain of dependent global reads;
//   2. the workgroup stages the small bodies (mapping tables, QCFG, the two
//      serialised trees) in LDS, all five in one round of loads;
//   3. the two trees are recovered side by side (512 lanes each, see "Pre-order tree
//      recovery" above); lanes 128..415 parse the tables;
//   4. the verdict is the FIRST failure in the reference's order of checks;
//   5. the decode tables of BOTH streams are built side by side, 512 lanes each.
// ---------------------------------------------------------------------------
constexpr int kParseThreads = 1024;
constexpr int kParseHalf = kParseThreads / 2;   // lanes per stream in step 5
__device__ __forceinline__ void identity_test_words(const DecFrame *df, int l, uint32_t *dst);

__global__ __launch_bounds__(kParseThreads) void k_dec_parse(Geom g, DecWs ws, const uint8_t *packed,
                                                             size_t in_stride, const uint32_t *sizes) {
  __shared__ TreeAux aux[2][kMaxNodes + 1];
  __shared__ uint32_t s_nd[2][kMaxNodes + 1];
  __shared__ __attribute__((aligned(16))) uint32_t s_lut[2][1 << kLutBits];
  __shared__ uint32_t s_sub[2][kSubEntries];
  __shared__ __attribute__((aligned(16))) uint16_t s_heap[2][2 << kLutBits];
  __shared__ uint16_t s_symd[2][kMaxNodes + 1];
  __shared__ uint32_t s_nslow[2], s_slow_m[2][kMaxSlow], s_slow_off[2][kMaxSlow];
  // Staged chunk bodies: 0 LMAP, 1 LRES tree, 2 QCFG, 3 FMAP, 4 FRES tree.
  constexpr int kBodyBytes = kTreeStride + 16;
  __shared__ uint32_t s_buf[5][kBodyBytes / 4];
  __shared__ uint32_t s_off[5], s_sz[5];
  __shared__ int32_t s_nn[2];
  static_assert(sizeof(TreeScratch) <= sizeof(s_heap) && sizeof(TreeScratch) <= sizeof(s_lut),
                "the trees' scratch lives in the memory of the heap and the LUTs (unused until step 5)");
  // Verdict per check, in the reference's order (0 = passed / not reached).
  enum { cHead = 0, cLmap, cLresFind, cLresTree, cQcfg, cFmapFind, cFmap, cFresFind, cFresTree, cLeaf0, cLeaf1, cCount };
  __shared__ int s_chk[cCount];
  __shared__ int s_status;
  const int f = blockIdx.x, lane = threadIdx.x;   // lane = thread index in the workgroup
  const long long c_in = clock64();
  const uint8_t *p = packed + (size_t)f * in_stride;
  const uint32_t n = sizes[f];
  DecFrame *df = ws.frames + f;

  if (lane < cCount) s_chk[lane] = 0;
  if (lane < 5) { s_off[lane] = 0; s_sz[lane] = 0; }
  if (lane < 2) s_nn[lane] = 0;
  __syncthreads();

  if (lane == 0) {   // ---- 1: chunk headers, in file order
    uint32_t idx = 12, sz = 0;
    do {
      if (n < 12 || rd32(p) != 0x46464952u /*RIFF*/ || rd32(p + 4) + 8u != n ||
          rd32(p + 8) != 0x474d4948u /*HIMG*/) { s_chk[cHead] = fmt_err(1, 0); break; }
      if (!find_chunk(p, n, &idx, 0x544d5246u /*FRMT*/, &sz) || sz < 11 || p[idx] != 1) { s_chk[cHead] = fmt_err(2, 0); break; }
      const uint32_t w = rd32(p + idx + 1), h = rd32(p + idx + 5);
      const int c = p[idx + 9];
      df->ycbcr = (p[idx + 10] != 0 && c >= 3) ? 1 : 0;
      if ((int)w != g.W || (int)h != g.H || c != g.C) { s_chk[cHead] = kStGeom; break; }
      idx += sz;
      if (!find_chunk(p, n, &idx, 0x50414d4cu /*LMAP*/, &sz)) { s_chk[cHead] = fmt_err(3, 0); break; }
      s_off[0] = idx; s_sz[0] = sz;
      idx += sz;
      if (!find_chunk(p, n, &idx, 0x5345524cu /*LRES*/, &sz)) { s_chk[cLresFind] = fmt_err(4, 0); break; }
      df->s[0].chunk_end = idx + sz;
      s_off[1] = idx; s_sz[1] = sz;
      idx += sz;
      if (!find_chunk(p, n, &idx, 0x47464351u /*QCFG*/, &sz) || sz != (df->ycbcr ? 64u : 32u)) { s_chk[cQcfg] = fmt_err(5, 0); break; }
      s_off[2] = idx; s_sz[2] = sz;
      idx += sz;
      if (!find_chunk(p, n, &idx, 0x50414d46u /*FMAP*/, &sz)) { s_chk[cFmapFind] = fmt_err(6, 0); break; }
      s_off[3] = idx; s_sz[3] = sz;
      idx += sz;
      if (!find_chunk(p, n, &idx, 0x53455246u /*FRES*/, &sz)) { s_chk[cFresFind] = fmt_err(7, 0); break; }
      df->s[1].chunk_end = idx + sz;
      // Trap T2: the decoder derives use_blocks from the COMPRESSED size
      // (huffman_dec.cpp:215-219); UncompressBlock refuses when it is false (:265).
      // (The opt-in fixed mode derives use_blocks like the encoder does: rows > 1.)
      if (!g.fix_t2 && !((uint32_t)g.row_block < sz)) { s_chk[cFresFind] = fmt_err(7, 1); break; }
      s_off[4] = idx; s_sz[4] = sz;
    } while (0);
  }
  __syncthreads();

  // ---- 2: stage the bodies (a dependent global load costs ~1 us, an LDS read ~50 ns)
  for (int k = lane; k < 5 * kBodyBytes; k += kParseThreads) {
    const int b = k / kBodyBytes, j = k - b * kBodyBytes;
    const uint32_t cnt = s_sz[b] < (uint32_t)kTreeStride ? s_sz[b] : (uint32_t)kTreeStride;
    reinterpret_cast<uint8_t *>(s_buf[b])[j] = (uint32_t)j < cnt ? p[s_off[b] + j] : (uint8_t)0;
  }
  __syncthreads();

  // ---- 3: the two trees side by side, 512 lanes each: LRES (decoder.cpp:232) and FRES
  // (decoder.cpp:290), huffman_dec.cpp:152-229
  {
    const int s = lane / kParseHalf, l = lane - s * kParseHalf, b = s ? 4 : 1;
    TreeScratch *ts = s ? reinterpret_cast<TreeScratch *>(&s_lut[0][0]) : reinterpret_cast<TreeScratch *>(&s_heap[0][0]);
    const uint32_t *w = s_buf[b];
    const bool present = s_off[b] != 0;
    const uint32_t nbytes = !present ? 0u : s_sz[b] < (uint32_t)kTreeStride ? s_sz[b] : (uint32_t)kTreeStride;
    const uint32_t bit_end = 8u * nbytes;
    const int nw = (int)((bit_end + 31u) >> 5);
    // A1: the walk over one word from every entry state
    if (l < nw) {
      const uint32_t x = w[l];
      unsigned long long outs = 0, cnts = 0;
      for (uint32_t e = 0; e < 10; ++e) {
        uint32_t pos = e, c = 0;
        while (pos < 32u) { ++c; pos += ((x >> pos) & 1u) ? 10u : 1u; }
        outs |= (unsigned long long)(pos - 32u) << (4u * e);
        cnts |= (unsigned long long)c << (6u * e);
      }
      ts->outs[l] = outs; ts->cnts[l] = cnts;
    }
    if (l == 0) { ts->first_bad = kNodeCap; ts->kend = kNodeCap; ts->kdeep = kNodeCap; }
    for (int k = l; k < kNodeCap + 6; k += kParseHalf) { ts->second[k] = 0; ts->key[k] = 0; }
    __syncthreads();
    // A2: chain the words
    if (l == 0) {
      uint32_t state = 0, base = 0;
      for (int k = 0; k < nw; ++k) {
        ts->went[k] = state | (base << 4);
        base += (uint32_t)(ts->cnts[k] >> (6u * state)) & 63u;
        state = (uint32_t)(ts->outs[k] >> (4u * state)) & 15u;
      }
      ts->total = (int32_t)(base < (uint32_t)kNodeCap ? base : (uint32_t)kNodeCap);
    }
    __syncthreads();
    // A3: the nodes' start bits
    if (l < nw) {
      const uint32_t x = w[l], e = ts->went[l] & 15u;
      uint32_t k = ts->went[l] >> 4, pos = e;
      while (pos < 32u && k < (uint32_t)kNodeCap) {
        const uint32_t leaf = (x >> pos) & 1u;
        ts->pos[k++] = (uint16_t)((32u * (uint32_t)l + pos) | (leaf << 15));
        pos += leaf ? 10u : 1u;
      }
    }
    __syncthreads();
    // B1: open[] and where the tree ends
    const int total = ts->total;
    for (int k = l; k < kNodeCap; k += kParseHalf) {
      int16_t d = 0;
      if (k < total) {
        const uint32_t pp = ts->pos[k], leaf = pp >> 15, bp = pp & 0x7fffu;
        d = leaf ? (int16_t)-1 : (int16_t)1;
        // ReadBitChecked / ReadBitsChecked, huffman_dec.cpp:51-60, 94-106
        if (bp >= bit_end || (leaf && bp + 10u > bit_end)) atomicMin(&ts->first_bad, k);
      }
      ts->open[0][k] = d;
    }
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < kNodeCap; d <<= 1) {
      for (int k = l; k < kNodeCap; k += kParseHalf)
        ts->open[cur ^ 1][k] = (int16_t)(ts->open[cur][k] + (k >= d ? ts->open[cur][k - d] : 0));
      cur ^= 1;
      __syncthreads();
    }
    const int16_t *open = ts->open[cur];   // (stack depth after node k) - 1
    for (int k = l; k < total; k += kParseHalf) {
      if (open[k] == -1) atomicMin(&ts->kend, k);
      if (!(ts->pos[k] >> 15)) ts->key[k] = (int16_t)(open[k] + 1);   // >= 2 for a branch
    }
    __syncthreads();
    // The first node the reference's walk fails on (count >= kMaxNodes, or the bits run
    // out: huffman_dec.cpp:51-60), if any.
    const int kend = ts->kend, first_bad = ts->first_bad;
    const bool whole = kend < kMaxNodes && first_bad > kend;
    const int ka = whole ? kNodeCap : min(min(first_bad, total), kMaxNodes);
    const int nlim = whole ? kend + 1 : ka;   // nodes the walk has read
    // B2: parents
    for (int k = l; k < nlim; k += kParseHalf) {
      int par = 0, which = 0;
      if (k > 0) {
        if (!(ts->pos[k - 1] >> 15)) {
          par = k - 1;
        } else {
          which = 1;
          const uint32_t t = (uint32_t)(uint16_t)(open[k - 1] + 2), tt = t | (t << 16);
          const int j0 = k - 2;
          int found = -1;
          for (int c = j0 >> 3; c >= 0 && found < 0; --c) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&ts->key[8 * c]);
            const uint32_t xs[4] = {q.x ^ tt, q.y ^ tt, q.z ^ tt, q.w ^ tt};
            uint32_t any = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) any |= (xs[i] - 0x00010001u) & ~xs[i] & 0x80008000u;
            if (!any) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int idx = 8 * c + i;
              if (idx <= j0 && ((xs[i >> 1] >> (16 * (i & 1))) & 0xffffu) == 0u) found = idx;
            }
          }
          par = found < 0 ? 0 : found;
          ts->second[par] = (uint16_t)k;
        }
      }
      ts->anc[k] = (uint16_t)par;
      TreeAux a; a.code = (uint32_t)which; a.depth = k > 0 ? 1 : 0;
      aux[s][k] = a;
    }
    __syncthreads();
    // B3: depth and code by pointer jumping (a depth beyond 63 stays at >= 32: rejected below)
    for (int r = 0; r < 6; ++r) {
      static_assert(kNodeCap <= 2 * kParseHalf, "two nodes per lane");
      const int k0 = l, k1 = l + kParseHalf;
      TreeAux n0, n1;
      uint16_t a0 = 0, a1 = 0;
      n0.code = n1.code = 0; n0.depth = n1.depth = 0;
      if (k0 < nlim) {
        const int a = ts->anc[k0];
        const TreeAux me = aux[s][k0], up = aux[s][a];
        n0.code = up.code | (up.depth < 32 ? me.code << up.depth : 0u);
        n0.depth = min(me.depth + up.depth, 255);
        a0 = ts->anc[a];
      }
      if (k1 < nlim) {
        const int a = ts->anc[k1];
        const TreeAux me = aux[s][k1], up = aux[s][a];
        n1.code = up.code | (up.depth < 32 ? me.code << up.depth : 0u);
        n1.depth = min(me.depth + up.depth, 255);
        a1 = ts->anc[a];
      }
      __syncthreads();
      if (k0 < nlim) { aux[s][k0] = n0; ts->anc[k0] = a0; }
      if (k1 < nlim) { aux[s][k1] = n1; ts->anc[k1] = a1; }
      __syncthreads();
    }
    // B4: a branch deeper than the decoder walks; the packed nodes
    for (int k = l; k < nlim; k += kParseHalf) {
      const uint32_t pp = ts->pos[k], leaf = pp >> 15;
      if (!leaf) {
        if (aux[s][k].depth + 1 > kMaxDepth) atomicMin(&ts->kdeep, k);
        s_nd[s][k] = ((uint32_t)(k + 1) & 1023u) | ((uint32_t)ts->second[k] << 10);
      } else {
        const uint32_t bp = (pp & 0x7fffu) + 1u, wi = bp >> 5;
        const uint32_t sym = (uint32_t)((((unsigned long long)w[wi + 1] << 32) | w[wi]) >> (bp & 31u)) & 511u;
        s_nd[s][k] = 0xfffffu | ((sym + 1u) << 20);
      }
    }
    __syncthreads();
    if (l == 0 && present) {
      // The first failure in the order of the reference's walk.
      int st = 0;
      if (ts->kdeep < ka) st = kStUnsupported;
      else if (!whole) st = fmt_err(s ? 7 : 4, 1);
      if (!st) {
        const int nn = kend + 1;
        const uint32_t tb = ((ts->pos[kend] & 0x7fffu) + 10u + 7u) >> 3;   // AlignToByte, huffman_dec.cpp:229
        df->s[s].num_nodes = nn;
        s_nn[s] = nn;
        df->s[s].root = 0;
        df->s[s].payload_off = s_off[b] + tb;
        // UncompressStream's first test (huffman_dec.cpp:277-278): nothing left after the tree.
        if (df->s[s].payload_off >= df->s[s].chunk_end) st = fmt_err(s ? 7 : 4, 1);
        // A tree that is a single leaf decodes without consuming code bits in the
        // reference (huffman_dec.cpp:173-185 with bits == 0) and cannot round-trip
        // the encoder's 1-bit codes; such streams are rejected.
        // (Fixed mode: read them as the 1-bit codes the encoder writes, huffman_enc.cpp:231-237.)
        if (!st && nn == 1 && !g.fix_t2) s_chk[s ? cLeaf1 : cLeaf0] = fmt_err(s ? 7 : 4, 1);
      }
      s_chk[s ? cFresTree : cLresTree] = st;
    }
  }
  if (lane >= 128 && lane < 384) {       // mapping tables: one entry per lane
    const int m = (lane - 128) >> 7, i = (lane - 128) & 127, b = m ? 3 : 0;
    if (s_off[b]) {
      const uint8_t *in = reinterpret_cast<const uint8_t *>(s_buf[b]);
      const bool ok = s_sz[b] <= (uint32_t)kTreeStride && map_ok(in, s_sz[b]);
      if (ok) (m ? df->fmap : df->lmap)[i] = map_entry(in, i);
      else if (i == 0) s_chk[m ? cFmap : cLmap] = fmt_err(m ? 6 : 3, 0);
    }
  } else if (lane >= 384 && lane < 416) {   // QCFG, quantize.cpp:190-213
    if (s_off[2]) {
      const uint8_t *q = reinterpret_cast<const uint8_t *>(s_buf[2]);
      const int i = lane - 384;
      df->shift[0][2 * i] = q[i] >> 4; df->shift[0][2 * i + 1] = q[i] & 15;
      const uint8_t x = df->ycbcr ? q[32 + i] : 0;
      df->shift[1][2 * i] = x >> 4; df->shift[1][2 * i + 1] = x & 15;
    }
  }
  __syncthreads();
  if (lane == 0) {   // ---- 4: first failure in the reference's order
    int st = 0;
    for (int k = 0; k < cCount && !st; ++k) st = s_chk[k];
    s_status = st;
    df->status = st;
    df->parse_status = st;
  }
  __syncthreads();
  if (s_status) return;
  // The row kernels' tables (DecFrame::row_tabs).  df->fmap / df->shift were written by other
  // lanes of this workgroup in front of the two barriers above.
  {
    int16_t *um = reinterpret_cast<int16_t *>(df->row_tabs);
    uint8_t *sh = reinterpret_cast<uint8_t *>(df->row_tabs + 128);
    uint32_t *sp = df->row_tabs + 160;
    if (lane < 256) {
      const int sc = (int8_t)lane;
      um[lane] = (int16_t)(sc >= 0 ? df->fmap[sc] : (sc == -128 ? -df->fmap[127] : -df->fmap[-sc]));
    } else if (lane < 384) {
      sh[lane - 256] = df->shift[(lane - 256) >> 6][(lane - 256) & 63];
    } else if (lane < 448) {
      // The same shifts as packed pairs in tile_plane's register order.
      const int t = lane - 384, ch = t >> 5, e = t & 31, x = e >> 2, j = e & 3;
      sp[t] = (uint32_t)df->shift[ch][(2 * j) * 8 + x] | ((uint32_t)df->shift[ch][(2 * j + 1) * 8 + x] << 16);
    } else if (lane < 512) {
      identity_test_words(df, lane - 448, sp + 64);
    }
  }
  const long long c_serial = clock64();

  // ---- 5: both streams side by side
  const int s = lane / kParseHalf, l = lane - s * kParseHalf;
  const int nn = s_nn[s];
  // The packed nodes, for the tree walks of the row kernels (codes longer than both tables).
  {
    uint32_t *nodes = ws.nodes + ((size_t)f * 2 + s) * (kMaxNodes + 1);
    for (int k = l; k < nn; k += kParseHalf) nodes[k] = s_nd[s][k];
  }
  // First-level LUTs (kLutBits wide, LSB-first codes index them directly) in LDS,
  // then the group tables derived from them.  Every node of depth <= kLutBits is
  // first entered in an implicit heap (index (1 << depth) | code); an entry then
  // finds its leaf with at most kLutBits independent reads instead of every leaf
  // replicating itself (a depth-1 leaf would write half the table on one lane).
  static_assert(sizeof(s_heap) == 16 * kParseThreads, "one 16-byte store per lane");
  reinterpret_cast<uint4 *>(&s_heap[0][0])[lane] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (int k = l; k < nn; k += kParseHalf) {
    const int depth = aux[s][k].depth;
    if (depth > kLutBits) continue;
    const int sym = (int)(s_nd[s][k] >> 20) - 1;
    // leaf: k + 1; branch (only looked at on the last level): 0x8000 | k
    s_heap[s][(1u << depth) | aux[s][k].code] = (uint16_t)(sym >= 0 ? k + 1 : (0x8000 | k));
    s_symd[s][k] = (uint16_t)(sym >= 0 ? sym : 0x8000);
  }
  __syncthreads();
  for (uint32_t idx = l; idx < (1u << kLutBits); idx += kParseHalf) {
    uint32_t e = 0;
    // depth 0: a tree that is one leaf (rejected above, kept consistent anyway)
    uint32_t h = s_heap[s][1];
    if (h && !(h & 0x8000u)) e = lut_leaf(s_symd[s][h - 1], g.fix_t2 ? 1 : 0) | ((h - 1) << 20);
    for (int d = 1; d <= kLutBits && !e; ++d) {
      h = s_heap[s][(1u << d) | (idx & ((1u << d) - 1u))];
      if (h && !(h & 0x8000u)) e = lut_leaf(s_symd[s][h - 1], d) | ((h - 1) << 20);
      else if (h && d == kLutBits) e = lut_node(h & 0x7fffu, d);
    }
    s_lut[s][idx] = e;
  }
  if (l == 0) s_nslow[s] = 0;
  for (int k = l; k < kSubEntries; k += kParseHalf) s_sub[s][k] = 0;
  __syncthreads();
  const long long c_lut = clock64();
  // Second-level tables for codes longer than kLutBits.  Few kLutBits-bit prefixes
  // lead to such codes; each gets a sub-table indexed by the next m bits (m = the
  // deepest leaf below it, at most kSubMaxBits; deeper leaves continue from a
  // branch node).  The flagged first-level entry keeps its slot in bits [8:0].
  for (uint32_t idx = l; idx < (1u << kLutBits); idx += kParseHalf) {
    const uint32_t e = s_lut[s][idx];
    if (e & 512u) {
      const uint32_t slot = atomicAdd(&s_nslow[s], 1u);
      s_lut[s][idx] = e | (slot < (uint32_t)kMaxSlow ? slot : 511u);
      if (slot < (uint32_t)kMaxSlow) s_slow_m[s][slot] = 0;
    }
  }
  __syncthreads();
  for (int k = l; k < nn; k += kParseHalf) {
    const int depth = aux[s][k].depth;
    if ((s_nd[s][k] >> 20) != 0 && depth > kLutBits) {
      const uint32_t slot = s_lut[s][aux[s][k].code & ((1u << kLutBits) - 1)] & 511u;
      if (slot < (uint32_t)kMaxSlow) atomicMax(&s_slow_m[s][slot], (uint32_t)(depth - kLutBits));
    }
  }
  __syncthreads();
  if (l == 0) {
    const uint32_t ns = s_nslow[s] < (uint32_t)kMaxSlow ? s_nslow[s] : (uint32_t)kMaxSlow;
    uint32_t off = 0;
    for (uint32_t q = 0; q < ns; ++q) {
      uint32_t m = s_slow_m[s][q] < (uint32_t)kSubMaxBits ? s_slow_m[s][q] : (uint32_t)kSubMaxBits;
      if (off + (1u << m) > (uint32_t)kSubEntries) m = 0;   // no room: plain tree walk
      s_slow_m[s][q] = m;
      s_slow_off[s][q] = off;
      if (m) off += 1u << m;
    }
  }
  __syncthreads();
  for (int k = l; k < nn; k += kParseHalf) {
    const int depth = aux[s][k].depth;
    if (depth <= kLutBits) continue;
    const uint32_t code = aux[s][k].code;
    const uint32_t slot = s_lut[s][code & ((1u << kLutBits) - 1)] & 511u;
    if (slot >= (uint32_t)kMaxSlow) continue;
    const uint32_t m = s_slow_m[s][slot], off = s_slow_off[s][slot];
    const uint32_t d = (uint32_t)(depth - kLutBits), rel = code >> kLutBits;
    const int sym = (int)(s_nd[s][k] >> 20) - 1;
    if (m == 0 || d > m) continue;
    if (sym >= 0) {
      // A valid leaf: the token resolved in the decoder's step format.  A symbol the
      // reference rejects stays a node reference: the walk ends on it and flags it.
      const uint32_t e = sym <= 260 ? sub_leaf(sym, d) : lut_node(k, depth);
      for (uint32_t i = 0; i < (1u << (m - d)); ++i) s_sub[s][off + ((i << d) | rel)] = e;
    } else if (d == m) {
      s_sub[s][off + rel] = lut_node(k, depth);
    }
  }
  __syncthreads();
  {
    // Out in the decoder's entry form: .x the literal byte, .y the step word of the ONE
    // token (code bits = what is left of the code after the kLutBits-bit prefix); a node
    // reference as .x = node | depth << 16, .y = 0 (also "nothing there": 0, 0).
    uint2 *sub = ws.sub + ((size_t)f * 2 + s) * kSubEntries;
    for (int k = l; k < kSubEntries; k += kParseHalf) {
      const uint32_t e = s_sub[s][k];
      uint2 o;
      if (e & 31u) {
        const uint32_t rest = e & 31u, eb = (e >> 5) & 31u, cb = (e >> 10) & 511u;
        const uint32_t cls = eb == 0 ? (cb == 2 ? 1u : 0u) : (eb == 2 ? 2u : eb == 4 ? 3u : eb == 8 ? 4u : 5u);
        o.x = (e >> 19) & 255u;
        o.y = grp_y(rest, eb, cb, rest, cls);
      } else {
        o.x = (e >> 20) | (((e >> 10) & 63u) << 16);
        o.y = 0;
      }
      sub[k] = o;
    }
  }
  const long long c_sub = clock64();
  {
    // Group table of the lean decoder (GrpTables below).  A group is a greedy
    // sequence of tokens whose CODES lie inside the kLutBits known bits:
    // literals / single zeros / the two-zeros symbol for at most 4 output bytes,
    // optionally closed by ONE zero-run token (its zeros need no explicit bytes;
    // its extra bits are read from the stream at decode time).
    uint2 *grp = ws.grp + ((size_t)f * 2 + s) * (1u << kLutBits);
    for (uint32_t idx = l; idx < (1u << kLutBits); idx += kParseHalf) {
      uint32_t used = 0, nout = 0, bytes = 0, g_eb = 0, s_tb = 0, s_class = 0, ntok = 0;
      const uint32_t e0 = s_lut[s][idx];
      for (;;) {
        const uint32_t e = s_lut[s][(idx >> used) & ((1u << kLutBits) - 1)];
        const uint32_t len = (e >> 10) & 63u, sym = e & 511u;
        if ((e & 512u) || len == 0 || sym > 260 || used + len > (uint32_t)kLutBits) break;
        uint32_t cls;
        if (sym <= 256) {
          const uint32_t cnt = sym == 256 ? 2u : 1u;
          if (nout + cnt > 4) break;
          if (sym < 256) bytes |= sym << (8 * nout);
          nout += cnt;
          cls = sym == 256 ? 1u : 0u;
        } else {
          cls = sym - 255u;  // 2..5
          nout += cls == 2 ? 3u : cls == 3 ? 7u : cls == 4 ? 23u : 279u;
          g_eb = cls == 2 ? 2u : cls == 3 ? 4u : cls == 4 ? 8u : 14u;
        }
        used += len;
        if (ntok++ == 0) { s_tb = len; s_class = cls; }
        if (sym > 256) break;
      }
      uint2 r;
      if (ntok) {
        r.x = bytes;
        r.y = grp_y(used, g_eb, nout, s_tb, s_class);
      } else {
        // The first code is longer than the table: .x says where to continue --
        // bit 31 set: sub-table (offset << 8 | index bits); else the tree walk
        // from node | depth << 16.
        const uint32_t slot = e0 & 511u;
        if ((e0 & 512u) && slot < (uint32_t)kMaxSlow && s_slow_m[s][slot])
          r.x = 0x80000000u | (s_slow_off[s][slot] << 8) | s_slow_m[s][slot];
        else
          r.x = (e0 >> 20) | (((e0 >> 10) & 63u) << 16);
        r.y = 0;
      }
      grp[idx] = r;
      // The same greedy group WITHOUT the four-byte limit, for the passes that only
      // count (k_row_count, lead-ins): as many tokens as have their codes inside the
      // kLutBits known bits -- more bits per step where zeros and short literals
      // alternate.
      uint32_t yc = 0;
      if (ntok) {
        uint32_t u2 = 0, n2 = 0, eb2 = 0;
        for (;;) {
          const uint32_t e = s_lut[s][(idx >> u2) & ((1u << kLutBits) - 1)];
          const uint32_t len = (e >> 10) & 63u, sym = e & 511u;
          if ((e & 512u) || len == 0 || sym > 260 || u2 + len > (uint32_t)kLutBits) break;
          u2 += len;
          if (sym <= 256) { n2 += sym == 256 ? 2u : 1u; continue; }
          const uint32_t cls = sym - 255u;
          n2 += cls == 2 ? 3u : cls == 3 ? 7u : cls == 4 ? 23u : 279u;
          eb2 = cls == 2 ? 2u : cls == 3 ? 4u : cls == 4 ? 8u : 14u;
          break;
        }
        yc = grp_y(u2, eb2, n2, s_tb, s_class);
      }
      ws.gyc[((size_t)f * 2 + s) * (1u << kLutBits) + idx] = yc;
    }
  }
  if (lane == 0) {   // phase cycles / 16 for tools/dec_stats.py (after the memset of stats)
    uint32_t *st = ws.parse_stats + (size_t)f * 4;
    st[0] = (uint32_t)((c_serial - c_in) >> 4); st[1] = (uint32_t)((c_lut - c_serial) >> 4);
    st[2] = (uint32_t)((c_sub - c_lut) >> 4); st[3] = (uint32_t)((clock64() - c_sub) >> 4);
  }
}

// ---------------------------------------------------------------------------
// k_dec_rowwalk: index of the FRES block rows (huffman_dec.cpp:232-248).  Every
// row's size header sits right behind the previous row's payload, so this is a
// chain of dependent loads (0.45 us per hop for 4096-pixel rows, 0.64 us for
// 16384-pixel ones) that nothing can parallelise; it runs on a side stream, beside
// k_dec_parse and the LRES kernels.  (Tried and dropped, profiles/r03_experiments.md:
// other waves of the workgroup touching the chunk a bounded distance ahead so that the
// hops hit in L2 -- slower; the headers through the scalar cache -- 9 % faster at
// 4096 pixels, 38 % slower at 16384.)
// ---------------------------------------------------------------------------
// A single frame's walk is cut into row ranges, one launch each (row_end: walk the rows in
// front of it; the last launch walks to the end of the chunk; resume: continue where the
// launch before stopped, DecFrame::walk_q / walk_r): the counts and the row kernels of a
// range run while the next range is walked (launch_decode).
__global__ __launch_bounds__(64) void k_dec_rowwalk(Geom g, DecWs ws, const uint8_t *packed,
                                                    size_t in_stride, const uint32_t *sizes,
                                                    int row_end, int resume) {
  // The walk needs where the FRES payload starts and ends -- which k_dec_parse knows
  // only after its tree recovery.  It finds both by itself (the same chunk
  // look-ups, then only the LENGTH of the serialised tree: a leaf is 1 + 9 bits, a
  // branch 1 bit, pre-order, huffman_dec.cpp:152-229) and so runs beside k_dec_parse
  // instead of behind it.  Whatever is wrong with the headers or the tree is
  // k_dec_parse's to report; this kernel reports the row headers only, in
  // walk_status, which k_row_count / k_dec_status merge once both kernels are done.
  __shared__ uint32_t s_tree[(kTreeStride + 16) / 4];
  __shared__ uint32_t s_hdr[2];
  const int f = blockIdx.x, lane = threadIdx.x;
  DecFrame *df = ws.frames + f;
  const uint8_t *p = packed + (size_t)f * in_stride;
  uint32_t q = 0, end = 0;
  int r = 0;
  if (resume) {
    if (lane != 0) return;
    q = df->walk_q; r = (int)df->walk_r; end = df->walk_end;
    if (q == 0) return;   // finished (or never started): the verdict is in
  } else {
    const uint32_t n = sizes[f];
    if (lane == 0) {
      uint32_t idx = 12, sz = 0;
      bool ok = n >= 12;
      const uint32_t tags[6] = {0x544d5246u /*FRMT*/, 0x50414d4cu /*LMAP*/, 0x5345524cu /*LRES*/,
                                0x47464351u /*QCFG*/, 0x50414d46u /*FMAP*/, 0x53455246u /*FRES*/};
      for (int t = 0; ok && t < 6; ++t) {   // the order of k_dec_parse (decoder.cpp:144-290)
        ok = find_chunk(p, n, &idx, tags[t], &sz);
        if (ok && t < 5) idx += sz;
      }
      s_hdr[0] = ok ? idx : 0u;
      s_hdr[1] = ok ? sz : 0u;
      df->walk_status = 0;
      df->rows_first = 0;
      df->walk_q = 0;
    }
    __syncthreads();
    const uint32_t coff = s_hdr[0], csz = s_hdr[1];
    if (coff == 0) return;
    const uint32_t cnt = csz < (uint32_t)kTreeStride ? csz : (uint32_t)kTreeStride;
    for (uint32_t k = lane; k < (uint32_t)kTreeStride + 16u; k += 64u)
      reinterpret_cast<uint8_t *>(s_tree)[k] = k < cnt ? p[coff + k] : (uint8_t)0;
    __syncthreads();
    if (lane != 0) return;
    uint32_t bit = 0;
    {
      // Length of the serialised tree over a 64-bit register window.
      unsigned long long win = ((unsigned long long)s_tree[1] << 32) | s_tree[0];
      uint32_t next = 2, ahead = s_tree[2];
      const uint32_t bit_end = 8u * cnt;
      int open = 1, count = 0, nb = 64;
      while (open > 0) {
        if (count >= kMaxNodes || bit >= bit_end) return;   // k_dec_parse rejects this tree
        ++count;
        if (nb <= 32) { win |= (unsigned long long)ahead << nb; nb += 32; ahead = s_tree[++next]; }
        if (win & 1ull) {
          if (bit + 10u > bit_end) return;
          win >>= 10; nb -= 10; bit += 10u;
          --open;
        } else {
          win >>= 1; nb -= 1; bit += 1u;
          ++open;
        }
      }
    }
    q = coff + ((bit + 7u) >> 3);   // AlignToByte, huffman_dec.cpp:229
    end = coff + csz;
    if (q >= end) return;                    // nothing behind the tree: k_dec_parse's verdict
    df->rows_first = q;
    if (g.fix_t2 && g.rows == 1) {   // the encoder writes one block row without a size header
      ws.row_off[(size_t)f * g.rows] = q;
      ws.row_len[(size_t)f * g.rows] = end - q;
      return;
    }
  }
  uint32_t *ro = ws.row_off + (size_t)f * g.rows, *rl = ws.row_len + (size_t)f * g.rows;
  int st = 0;
  while (q != end && r < row_end) {
    if (q + 2 > end) { st = fmt_err(7, 1); break; }
    uint32_t len = p[q] | (p[q + 1] << 8);
    q += 2;
    if (len & 0x8000u) {
      if (q + 2 > end) { st = fmt_err(7, 1); break; }
      len = (len & 0x7fffu) | ((uint32_t)(p[q] | (p[q + 1] << 8)) << 15);
      q += 2;
    }
    if (len > end - q) { st = fmt_err(7, 1); break; }
    if (r < g.rows) { ro[r] = q; rl[r] = len; }
    ++r;
    q += len;
  }
  if (st || q == end) {
    if (!st && r < g.rows) st = fmt_err(7, 1);  // fewer blocks than block rows
    df->walk_status = st;
    df->walk_q = 0;
  } else {   // the next launch goes on from here
    df->walk_q = q; df->walk_r = (uint32_t)r; df->walk_end = end;
  }
}

// k_dec_set_index: the row index comes from the caller (row-sharded decode: the rank
// that holds the whole stream walked the headers once, the others hold only their own
// rows' bytes and cannot) -- [rows] payload offsets, then [rows] lengths.  Rows whose
// payload would leave the stream are flagged like a damaged header.
__global__ __launch_bounds__(256) void k_dec_set_index(Geom g, DecWs ws, const uint32_t *index,
                                                       const uint32_t *sizes, int r0, int r1) {
  DecFrame *df = ws.frames;   // (one frame)
  const uint32_t n = sizes[0];
  int bad = 0;
  for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
    const uint32_t off = index[r], len = index[g.rows + r];
    ws.row_off[r] = off;
    ws.row_len[r] = len;
    if (off > n || len > n - off) bad = 1;
  }
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) { df->walk_status = bad ? fmt_err(7, 1) : 0; df->rows_first = 0; }
}

// ---------------------------------------------------------------------------
// Entropy decoding (huffman_dec.cpp:274-418), parallel inside one stream.
//
// A whole workgroup (1024 lanes) decodes one Huffman stream (the LRES stream or
// one FRES block row).  The payload is processed in chunks of 1024 sub-sequences.
// Every lane decodes one sub-sequence speculatively from its nominal start; then
// the workgroup iterates start[t+1] = end[t] until nothing changes.  Lane 0's
// start is exact, so by induction the fixpoint is the exact token chain (Huffman
// streams self-synchronise, typically within one sub-sequence).  A prefix scan
// of the per-lane symbol counts places the output, and a last pass writes it.
//
// These passes are bound by VALU issue (a CU retires one wave64 VALU
// instruction per cycle), so the inner loop is kept to a few dozen instructions
// per step: a 64-bit bit window per lane over the payload in place (L2), ONE
// table read per step that resolves a whole group of tokens, and no per-token
// classification.
//
// Group table entry (uint2), indexed by the next kLutBits stream bits:
//   .x  the first 4 output bytes of the group (0 where the output is a zero)
//   .y  [4:0]   code bits of the whole group   [9:5]   extra bits that follow them
//       [18:10] symbols it produces, before the extra-bits value is added
//       [22:19] code bits of its first token    [25:23] class of the first token:
//               0 literal / single zero, 1 two zeros, 2..5 zero runs 257..260
//       [31:27] [4:0] + [9:5]: the bits the step consumes
//       (so the extra-bits value is v_bfe_u32(window, y, y >> 5) -- the hardware
//       takes the low five bits of either operand -- and the step is five
//       instructions from the table word to the new window)
//   .y == 0: the first code is longer than the table; .x says where to continue:
//            bit 31 set: second-level table (offset << 8 | index bits), whose
//            entries are step words of ONE token (sub_leaf) or node references;
//            else node | depth << 16 for the tree walk.
// A lane owns the tokens that START in [start, lim): the whole group is taken
// while pos + kLutBits <= lim (every token of it then starts before lim), the
// first token alone otherwise.
// ---------------------------------------------------------------------------
constexpr int kWinBytes = 32768;  // output window in LDS (streams that go to HBM)
constexpr uint32_t kMinSubBits = 128;  // shortest sub-sequence a lane decodes (short rows: more lanes busy)
constexpr int kPayWords = 9216 + 8;     // k_row_count: LDS words for a row's payload
constexpr uint32_t kPayPad = 4;         // dwords past the payload the reader may touch (window look-ahead)
constexpr uint32_t kJoinBits = 64;     // a moved lane re-joins its earlier decode this far past its nominal start

struct GrpTables {
  const uint2 *grp;           // LDS: 1 << kLutBits first-level entries, then kSubEntries
                              // second-level entries (.x bytes / descriptor, .y step word)
  const uint32_t *gx, *gy;    // the same two tables as two arrays, each (1 << kLutBits) +
                              // kSubEntries long, gx right behind gy (count-only kernels:
                              // the hot read is .y alone, and a stride-2 dword read of the
                              // interleaved table uses every other LDS bank only)
  const uint32_t *nd;         // LDS tree nodes: child a | child b << 10 | (symbol + 1) << 20
};
constexpr int kTabEntries = (1 << kLutBits) + kSubEntries;

// The decode tables of one stream as they sit in LDS (interleaved form).
struct __attribute__((aligned(16))) LdsTables {
  uint2 grp[kTabEntries];
  uint32_t nd[kMaxNodes + 1];
};
__device__ __forceinline__ GrpTables tables_of(const LdsTables *T) {
  GrpTables t;
  t.grp = T->grp; t.gx = nullptr; t.gy = nullptr; t.nd = T->nd;
  return t;
}

// LDS byte address of a pointer into LDS, and loads through such an address (the
// hot loops keep "which table, how many index bits" as an address and a mask).
typedef const __attribute__((address_space(3))) uint32_t *lds_u32p;
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ uint32_t lds_ld32(uint32_t a) { return *(lds_u32p)(uintptr_t)a; }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 lds_ld64(uint32_t a) {
  const u32x2 v = *(const __attribute__((address_space(3))) u32x2 *)(uintptr_t)a;
  return make_uint2(v.x, v.y);
}

// Exclusive scan of a 64-bit value over the 1024-thread workgroup.
__device__ __forceinline__ unsigned long long block_scan_u64(unsigned long long v,
                                                             unsigned long long *sm,
                                                             unsigned long long *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) sm[wave] = incl;
  __syncthreads();
  unsigned long long pre = 0, tot = 0;
  for (int w = 0; w < kDecThreads / 64; ++w) {
    if (w < wave) pre += sm[w];
    tot += sm[w];
  }
  __syncthreads();
  *total = tot;
  return pre + incl - v;
}

struct StreamShared {            // small LDS state of the stream decoder
  uint32_t nxt[kDecThreads + 1];
  unsigned long long sm64[kDecThreads / 64];
  unsigned long long endbit;     // bits consumed when the output became complete
  int flag;
  int err;
  uint32_t dbg[2];               // diagnostics: lanes that re-joined / re-decoded in rounds >= 2
};

// Bit window over the stream's dwords in global memory.  Word indices are
// clamped to the dword that holds the stream's last byte: bits past the end of
// the stream repeat that dword, which only a token that overruns the payload can
// see -- and such a stream is rejected whatever those bits are.
template <class PTR>
struct ReaderT {
  PTR w;               // dword-aligned window base (global memory, or LDS for a staged payload)
  uint32_t jmax;       // index of the stream's last dword
  unsigned long long win;
  int nb;              // valid bits in win
  uint32_t next;       // index of the word held in pre
  uint32_t pre;        // prefetched word: the load issued by one refill is first
                       // touched by the next one, so its latency hides behind ~3 steps.
                       // (It must land in `pre` untouched -- any move or mask of the
                       // loaded register makes the compiler wait for it on the spot.)
  // Window over the stream `p` (4-byte aligned, stream_size bytes) whose bit 0 is
  // the dword holding absolute stream bit abs_bit; returns abs_bit's offset in it.
  __device__ __forceinline__ uint32_t attach(const uint8_t *p, uint32_t stream_size,
                                             unsigned long long abs_bit) {
    const uint32_t gb = (uint32_t)(abs_bit >> 5) * 4u;
    w = (PTR)reinterpret_cast<const uint32_t *>(p + gb);
    jmax = ((stream_size - 1u) >> 2) - (gb >> 2);
    return (uint32_t)(abs_bit - 8ull * gb);
  }
  __device__ __forceinline__ uint32_t ld(uint32_t j) const { return w[j < jmax ? j : jmax]; }
  __device__ __forceinline__ void init(uint32_t pos) {
    const uint32_t j = pos >> 5, sh = pos & 31;
    win = (((unsigned long long)ld(j + 1) << 32) | ld(j)) >> sh;
    nb = 64 - (int)sh;
    next = j + 2;
    pre = ld(next);
  }
  __device__ __forceinline__ void refill() {   // afterwards nb >= 33
    if (nb <= 32) {
      win |= (unsigned long long)pre << nb;
      nb += 32;
      ++next;
      pre = ld(next);
    }
  }
  __device__ __forceinline__ void consume(int n) { win >>= n; nb -= n; }
  // End of a walk: the last prefetch is never used, and a load that is still in flight when
  // the walk's code is left stays "pending" on its register for the compiler's s_waitcnt
  // placement in whatever code follows -- including, through the structurised control flow,
  // loops of OTHER walks that can never run after this one: their ds_read into that register
  // then carries an s_waitcnt vmcnt(0) on every iteration (round 5: the chain loop of the row
  // kernel's write pass waited for its own prefetch on every step because the token-tail
  // loop's prefetch register was still pending on a static path into it).  Using the word
  // here makes the wait happen once, where the walk ends.
  __device__ __forceinline__ void retire() { asm volatile("" :: "v"(pre)); }
};
typedef ReaderT<const uint32_t *> GReader;                                        // payload in place (L2)
typedef ReaderT<const __attribute__((address_space(3))) uint32_t *> LReader;      // payload staged in LDS

// Words [0, n) of a global reader's window -> LDS (dst 16-byte aligned, room for n
// rounded up to 4): four words per lane and step with ONE 16-byte load (the stream is
// only dword aligned; global memory takes that), so a 33 KiB row is three loads per
// lane in flight instead of nine one after the other.  Words past the stream's last
// dword repeat it, exactly like ReaderT::ld, and nothing past it is read.
struct __attribute__((packed, aligned(4))) PackedU4 { uint32_t x, y, z, w; };
__device__ __forceinline__ void stage_payload(const GReader &rd, uint32_t *dst, uint32_t n) {
  for (uint32_t k = threadIdx.x; 4u * k < n; k += kDecThreads) {
    uint4 v;
    if (4u * k + 3u <= rd.jmax) {
      const PackedU4 q = *reinterpret_cast<const PackedU4 *>(rd.w + 4u * k);
      v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
    } else {
      v.x = rd.ld(4u * k); v.y = rd.ld(4u * k + 1u); v.z = rd.ld(4u * k + 2u); v.w = rd.ld(4u * k + 3u);
    }
    *reinterpret_cast<uint4 *>(dst + 4u * k) = v;
  }
}

// Extra bits and run base per token class (huffman_common.h:24-28).
__device__ __forceinline__ uint32_t class_eb(uint32_t c) { return (0xE84200u >> (4u * c)) & 15u; }
__device__ __forceinline__ uint32_t class_base(uint32_t c) {
  const unsigned long long kBases = 1ull | (2ull << 9) | (3ull << 18) | (7ull << 27) | (23ull << 36) | (279ull << 45);
  return (uint32_t)(kBases >> (9u * c)) & 511u;
}

// The step word of a group entry reduced to its FIRST token (a lane's last
// kLutBits bits: it owns the tokens that START in its range).
__device__ __forceinline__ uint32_t first_token_word(uint32_t y) {
  const uint32_t c = (y >> 23) & 7u, tb = (y >> 19) & 15u, eb = class_eb(c);
  return tb | (eb << 5) | (class_base(c) << 10) | ((tb + eb) << 27);
}

// The tree walk for whatever neither table resolves (huffman_dec.cpp:291-328): from
// the node and depth in x (node | depth << 16; 0: nothing there), `base` code bits of
// the token already consumed from the window.  Consumes the rest of the code, returns
// a step word whose code-bit count is 0 (only the extra bits are left to take);
// *len_out = the code's length, *byte = the literal (0 for runs).  bad is set (never
// cleared) on symbols the reference rejects (huffman_dec.cpp:349-352).
template <class RD>
__device__ __forceinline__ uint32_t walk_token(RD &rd, const GrpTables &t, uint32_t x, int base,
                                               uint32_t *len_out, uint32_t *byte, bool *bad) {
  int node = (int)(x & 0xffffu), len = (int)((x >> 16) & 0x7fffu);
  uint32_t nd = t.nd[node];
  while (len >= base && (nd >> 20) == 0 && len < kMaxDepth) {
    node = (int)(((rd.win >> (len - base)) & 1ull) ? (nd >> 10) & 1023u : nd & 1023u);
    nd = t.nd[node];
    ++len;
  }
  const int sym = (int)(nd >> 20) - 1;
  const bool ok = sym >= 0 && sym <= 260 && len > base;
  if (!ok) *bad = true;
  if (len <= base) len = base + 1;  // keep moving on a degenerate tree
  rd.consume(len - base);
  rd.refill();
  *len_out = (uint32_t)len;
  const uint32_t c = !ok ? 0u : (sym < 256 ? 0u : (uint32_t)sym - 255u);
  const uint32_t eb = class_eb(c);
  *byte = (ok && sym < 256) ? (uint32_t)sym : 0u;
  return (eb << 5) | ((ok ? class_base(c) : 0u) << 10) | (eb << 27);
}

// A token whose code is longer than the first-level table, resolved in ONE call (the
// tails and the exact paths; the hot loops spread it over two steps instead, see
// lean_count): second-level table, the tree below it for whatever is deeper still.
// Consumes the code bits and returns a step word whose code-bit count is 0.
// *pre = code bits consumed, *byte = the literal (0 for runs).
template <bool SOA, class RD>
__device__ __forceinline__ uint32_t long_token(RD &rd, const GrpTables &t, uint32_t idx, uint32_t *pre,
                                            uint32_t *byte, bool *bad) {
  uint32_t x = SOA ? t.gx[idx] : t.grp[idx].x;
  int base = 0;
  if (x >> 31) {
    rd.consume(kLutBits);
    rd.refill();
    base = kLutBits;
    const uint32_t j = (1u << kLutBits) + ((x >> 8) & 0xffffu) + __builtin_amdgcn_ubfe((uint32_t)rd.win, 0, x & 255u);
    const uint32_t y2 = SOA ? t.gy[j] : t.grp[j].y;
    x = SOA ? t.gx[j] : t.grp[j].x;
    if (y2) {   // resolved: the common case by far
      const uint32_t rest = y2 & 31u, eb = (y2 >> 5) & 31u;
      rd.consume((int)rest);
      rd.refill();
      *pre = kLutBits + rest;
      *byte = x;
      return (y2 & 0x0007ffe0u) | (eb << 27);
    }
  }
  return walk_token(rd, t, x, base, pre, byte, bad);
}

// One decode step in its general form (the tails and the exact paths; the hot
// loops below inline the same thing without the `single` test).
template <bool WANT_BYTES, bool SOA = false, class RD = GReader>
__device__ __forceinline__ void lean_step(RD &rd, const GrpTables &t, bool single,
                                          uint32_t *nbits, uint32_t *count, uint32_t *bytes,
                                          bool *bad) {
  rd.refill();
  const uint32_t idx = (uint32_t)rd.win & ((1u << kLutBits) - 1u);
  uint32_t y, by = 0, pre = 0;
  if (WANT_BYTES) { const uint2 e = t.grp[idx]; by = e.x; y = e.y; }
  else if (SOA) y = t.gy[idx];
  else y = reinterpret_cast<const uint32_t *>(t.grp)[2 * idx + 1];
  if (__builtin_expect(y == 0, 0)) {
    y = long_token<SOA>(rd, t, idx, &pre, &by, bad);
  } else if (single) {
    y = first_token_word(y);
    by &= 255u;
  }
  const uint32_t extra = __builtin_amdgcn_ubfe((uint32_t)rd.win, y, y >> 5);   // the hardware takes [4:0] of either
  const uint32_t n = y >> 27;
  rd.consume((int)n);
  *nbits = pre + n;
  *count = ((y >> 10) & 511u) + extra;
  *bytes = by;
}

// Count pass: the tokens that start in [pos, lim).  Returns where the last one
// ends and how many symbols they produce.  Two loops: whole groups while every
// token of a group is sure to start before lim (pos + kLutBits <= lim), single
// tokens for the last few bits -- so that the hot loop carries no `single` test.
// cont: the reader already stands at pos (a previous call ended there).
// GRP: the lane owns the GROUPS that start in [pos, lim) -- one loop, no single-token
// tail; the walk ends at the first group boundary at or past lim (up to kLutBits + 14
// bits behind it).  Any token boundary divides two lanes' ranges correctly as long as
// the consumer walks from its own start to its right neighbour's start (the row kernels
// do), so the count kernels need not agree on a canonical one.
template <bool SOA = false, class RD = GReader, bool GRP = false>
__device__ __forceinline__ void lean_count(RD &rd, const GrpTables &t, uint32_t pos,
                                           uint32_t lim, uint32_t *endpos, uint32_t *count,
                                           bool cont = false) {
  uint32_t c = 0;
  if (pos < lim) {
    if (!cont) rd.init(pos);
    const int limk = (int)lim - kLutBits;
    bool bad = false;
    // The table a step indexes is lane state (tm: index mask, tb: LDS address, both in
    // bytes): a code longer than the first-level table is NOT resolved on the spot --
    // that was a 40-instruction detour with two dependent LDS reads which the whole
    // wavefront waited for whenever one of its 64 lanes met such a code (57 % of the
    // steps) -- but taken as a step that consumes the kLutBits known bits, produces
    // nothing and points the lane's NEXT step at the code's second-level table.
    constexpr int SH = SOA ? 2 : 3;
    const uint32_t TM = ((1u << kLutBits) - 1u) << SH;
    const uint32_t TB = SOA ? lds_addr(t.gy) : lds_addr(t.grp) + 4u;   // the .y words
    const uint32_t XOFF = SOA ? (uint32_t)kTabEntries * 4u : 0xfffffffcu;   // from .y to its .x
    uint32_t tm = TM, tb = TB;
    auto step = [&]() {
      rd.refill();
      const uint32_t a = ((((uint32_t)rd.win) << SH) & tm) + tb;
      uint32_t y = lds_ld32(a);
      uint32_t ntm = TM, ntb = TB, adv = 0;
      if (__builtin_expect(y == 0, 0)) {
        const uint32_t x = lds_ld32(a + XOFF);
        if ((x >> 31) && tm == TM) {
          ntm = ((1u << (x & 255u)) - 1u) << SH;
          ntb = TB + (((1u << kLutBits) + ((x >> 8) & 0xffffu)) << SH);
          y = (uint32_t)kLutBits | ((uint32_t)kLutBits << 27);
        } else {
          const int base = tm == TM ? 0 : kLutBits;
          uint32_t len, by;
          y = walk_token(rd, t, x, base, &len, &by, &bad);
          adv = len - (uint32_t)base;
        }
      }
      tm = ntm; tb = ntb;
      const uint32_t extra = __builtin_amdgcn_ubfe((uint32_t)rd.win, y, y >> 5);
      const uint32_t n = y >> 27;
      rd.consume((int)n);
      pos += adv + n;
      c += ((y >> 10) & 511u) + extra;
    };
    if constexpr (GRP) {
      while (pos < lim) step();
      if (tm != TM) step();
    } else {
      while ((int)pos <= limk) step();
      if (tm != TM) step();   // the loop ended between the two steps of a long code
      while (pos < lim) {
        uint32_t nbits, cnt, by;
        lean_step<false, SOA>(rd, t, true, &nbits, &cnt, &by, &bad);
        pos += nbits;
        c += cnt;
      }
    }
  }
  rd.retire();
  *endpos = pos;
  *count = c;
}

// Speculative decode of one chunk to the self-synchronised fixpoint.  On return
// lane t owns exactly the tokens that START in [*start, lim) -- given that lane
// 0's start `first` is exact (asserted or assumed by the caller).
// Cold: *start = nominal start, the lane decodes in round 1 iff `active`.
// Warm: (*start, *endpos, *cnt) hold a fixpoint reached for another `first`; only
// lanes whose start changes decode again.
// Lanes whose range lies beyond the payload (`active` false) own nothing and stay
// out of it: passing the chain's end along them would cost one round per lane.
// MEMO > 0: the lane remembers its last MEMO (start -> end, count) results in
// memo[3 * MEMO] (start ~0u = empty; the caller may pre-load it) and a start it has
// seen before costs a look-up.  For streams that do NOT self-synchronise -- codes of
// nearly one length, e.g. the 5/6-bit codes of the LRES predictor bytes: a wrong
// phase survives thousands of tokens there -- every round moves every lane of such a
// stretch to one of the same few phases again, and the correction passes along the
// chain one lane per round: rounds that are look-ups instead of decodes.
// NQ > 0: the lane's walk is cut at NQ more marks (mark[k], non-decreasing, between the
// re-join point and lim): on return mark[k] = the first token boundary at or past the mark
// and mcnt[k] = the symbols of the lane's tokens in front of it (k_row_count for rows that
// go through windows: four records per lane).
template <bool SOA = false, class RD = GReader, int MEMO = 0, int NQ = 0>
__device__ __forceinline__ void lean_fixpoint(RD &rd, const GrpTables &tb, StreamShared *sh,
                                              uint32_t first, bool active, uint32_t lim,
                                              uint32_t *start_io, uint32_t *endpos_io,
                                              uint32_t *cnt_io, uint32_t *rounds, bool warm,
                                              uint32_t lead_bits, long long *c_first = nullptr,
                                              long long *c_phase = nullptr, uint32_t *memo = nullptr,
                                              uint32_t *mark = nullptr, uint32_t *mcnt = nullptr) {
  const int tid = threadIdx.x;
  const long long t_in = clock64();
  uint32_t start = *start_io, endpos = *endpos_io, cnt = *cnt_io;
  const uint32_t nominal = start;   // cold: the lane's nominal boundary
  bool dirty = active;
  if (tid == 0 && !warm) start = first;
  // Lead-in: instead of starting blind at its nominal boundary, a lane decodes the
  // last lead_bits of its left neighbour's range first.  Huffman codes resynchronise
  // within a few tokens, so the first token boundary at or past the nominal start
  // found that way is almost always the true one, and round 1 is then already the
  // final round (1 + lead_bits/sub passes over the payload instead of 2+).  A lane
  // that did not synchronise is corrected by the rounds below as before.
  if (!warm && tid > 0 && active && lead_bits) {
    const uint32_t from = start - first > lead_bits ? start - lead_bits : first;
    uint32_t guess, none;
    lean_count<SOA>(rd, tb, from, start, &guess, &none);
    start = guess;
  }
  if (c_phase) c_phase[0] = clock64() - t_in;   // lead-in (this wave)
  if (warm) {
    dirty = (tid == 0) && (first != start);
    if (dirty) start = first;
  }
  // Re-join: a lane's decode is cut at T = nominal start + kJoinBits.  It remembers
  // the first token boundary at or past T (canonical: whatever the grouping, the
  // group that crosses T ends with a token that started before T) and the symbols
  // up to it.  When its start moves in a later round it decodes up to T again; if it
  // arrives at the same boundary, everything behind is what it already has -- the
  // end stays, the count is adjusted -- and the round costs kJoinBits instead of the
  // whole sub-sequence (codes resynchronise within a few tokens, so almost every
  // moved lane re-joins).
  uint32_t T = nominal + kJoinBits;
  if (T > lim || T < nominal) T = lim;
  uint32_t posT = ~0u, cT = 0;
  uint32_t mk[NQ > 0 ? NQ : 1], mp[NQ > 0 ? NQ : 1], mc[NQ > 0 ? NQ : 1];
  if (NQ > 0) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      uint32_t m = mark[k];
      if (m < T) m = T;
      if (m > lim) m = lim;
      mk[k] = m; mp[k] = start; mc[k] = 0;
    }
  }
  for (;;) {
    bool hit = false;
    if (MEMO > 0 && dirty) {
#pragma unroll
      for (int e = 0; e < MEMO; ++e)
        if (memo[3 * e] == start) { endpos = memo[3 * e + 1]; cnt = memo[3 * e + 2]; hit = true; }
      if (hit) posT = ~0u;   // (endpos, cnt) no longer belong to the re-join checkpoint
    }
    if (dirty && !hit) {
      uint32_t p1, c1;
      lean_count<SOA>(rd, tb, start, T, &p1, &c1);
      if (p1 == posT) {
        cnt = c1 + (cnt - cT);
        if (NQ > 0) {
#pragma unroll
          for (int k = 0; k < NQ; ++k) mc[k] += c1 - cT;
        }
        if (*rounds) atomicAdd(&sh->dbg[0], 1u);
      } else {
        if (*rounds) atomicAdd(&sh->dbg[1], 1u);
        uint32_t c2, pos = p1, acc = c1;
        bool at = start < T;   // the reader stands at `pos`
        if (NQ > 0) {
#pragma unroll
          for (int k = 0; k < NQ; ++k) {
            uint32_t ck;
            lean_count<SOA>(rd, tb, pos, mk[k], &mp[k], &ck, at);
            at = at || pos < mk[k];
            pos = mp[k];
            acc += ck;
            mc[k] = acc;
          }
        }
        lean_count<SOA>(rd, tb, pos, lim, &endpos, &c2, at);
        cnt = acc + c2;
      }
      posT = p1;
      cT = c1;
      if (MEMO > 0) {
#pragma unroll
        for (int e = MEMO - 1; e > 0; --e) {
          memo[3 * e] = memo[3 * e - 3]; memo[3 * e + 1] = memo[3 * e - 2]; memo[3 * e + 2] = memo[3 * e - 1];
        }
        memo[0] = start; memo[1] = endpos; memo[2] = cnt;
      }
    }
    sh->nxt[tid + 1] = endpos;
    __syncthreads();
    if (c_first && *c_first == 0) *c_first = clock64() - t_in;
    if (c_phase && c_phase[1] == 0) c_phase[1] = clock64() - t_in;   // end of round 1 (after its barrier)
    const uint32_t ns = tid == 0 ? first : sh->nxt[tid];
    dirty = active && (ns != start);
    if (active) start = ns;
    ++*rounds;
    if (!__syncthreads_or(dirty ? 1 : 0)) break;
  }
  *start_io = start;
  *endpos_io = endpos;
  *cnt_io = cnt;
  if (NQ > 0) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) { mark[k] = active ? mp[k] : start; mcnt[k] = active ? mc[k] : 0u; }
  }
}

// Write pass of one lane straight into LDS: the tokens that start in [bp, lim)
// go to lds_out (pre-zeroed) from offset op.  The group's bytes are OR-ed in with
// two aligned ds_or (the bytes past the group are zeros, so neighbours are never
// disturbed).  The caller guarantees that all of the lane's symbols lie strictly
// inside the block.  Same two loops as lean_count.
// CLIP: lds_out is a WINDOW of the block (k_row_window): op counts from kWinGuard
// bytes in front of the window's first symbol and the lane's symbols may lie before,
// inside or behind it; a group is OR-ed in when its first byte is less than 8 bytes in
// front of the window or inside it (win_span = window bytes + kWinGuard: one unsigned
// compare), so the two dwords it touches stay inside [0, win_span + 8).
constexpr uint32_t kWinGuard = 16;
constexpr uint32_t kRowWindow = 128u * 1024u;   // k_row_window: symbols of a row per workgroup (64 KiB windows, two
                                                // workgroups per CU: 1.12 -> 1.69 ms at 16384^2 -- half the lanes idle,
                                                // the per-workgroup costs twice)
// The chain form of lean_write (the reader stands at bp): [bp, lim) is a stretch of
// k_row_count_w's chain of groups, the groups taken from bp end exactly at lim.  The loop is bound
// by its VALU instructions (four wavefronts per SIMD; an instruction more per step is 1 % of the
// row kernel), scalar ones are free beside them.  Against lean_write's step:
//  * the verdict is a lane VALUE that only the tree walk touches (as a bool it lived in scalar
//    mask registers, three scalar instructions per step for a flag no valid stream sets);
//  * the reader's window base is the wavefront's (a row's payload): scalar base + a lane BYTE
//    offset -- the refill's address is one add and one min, the load takes the offset as it is;
//  * the output position carries the LDS address of the symbol area: the group's dword address is
//    one AND;
//  * a tree walk's extra bits advance the position where they are found, not through a register
//    that is zero on every other step.
struct ChainReader {
  unsigned long long win;
  int nb;
  uint32_t pre, noff, offmax;   // the prefetched word, its byte offset, the offset of the stream's last dword
  const __attribute__((address_space(1))) uint8_t *sbase;   // (wavefront-uniform; global)
  __device__ __forceinline__ void refill() {   // afterwards nb >= 33
    if (nb <= 32) {
      win |= (unsigned long long)pre << nb;
      nb += 32;
      noff += 4u;
      pre = *(const __attribute__((address_space(1))) uint32_t *)(sbase + min(noff, offmax));
    }
  }
  __device__ __forceinline__ void consume(int n) { win >>= n; nb -= n; }
};
__device__ __forceinline__ void lds_or32(uint32_t a, uint32_t v) {
  (void)__hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t *)(uintptr_t)a, v, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_WORKGROUP);
}
// CLIP (k_row_window): as in lean_write -- lds_out is a window of the block, a group is OR-ed in when its first byte
// lies less than 8 bytes in front of the window or inside it.
template <bool CLIP = false>
__device__ __forceinline__ bool lean_write_chain(GReader &rd0, const GrpTables &t, uint32_t bp, uint32_t lim, uint32_t op,
                                                 uint8_t *lds_out, uint32_t win_span = 0) {
  ChainReader rd;
  rd.win = rd0.win; rd.nb = rd0.nb; rd.pre = rd0.pre;
  rd.noff = 4u * rd0.next; rd.offmax = 4u * rd0.jmax;
  {
    const unsigned long long a = (unsigned long long)(uintptr_t)rd0.w;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    rd.sbase = (const __attribute__((address_space(1))) uint8_t *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
  }
  uint32_t badv = 0;
  uint32_t opa = op + lds_addr(lds_out);   // (the symbol area is dword aligned: opa & 3 == op & 3)
  const uint32_t clip_lo = lds_addr(lds_out) + (kWinGuard - 8u), clip_span = win_span - (kWinGuard - 8u);   // (CLIP)
  // (the table a step indexes is lane state: index bits and base -- v_bfe, v_lshl_add)
  const uint32_t TW = (uint32_t)kLutBits, TB = lds_addr(t.grp);
  uint32_t tm = TW, tb = TB;
  auto step = [&]() {
    rd.refill();
    const uint2 e = lds_ld64((__builtin_amdgcn_ubfe((uint32_t)rd.win, 0u, tm) << 3) + tb);
    uint32_t y = e.y, by = e.x, ntm = TW, ntb = TB;
    if (__builtin_expect(y == 0, 0)) {
      if ((by >> 31) && tm == TW) {
        ntm = by & 255u;
        ntb = TB + (((1u << kLutBits) + ((by >> 8) & 0xffffu)) << 3);
        y = (uint32_t)kLutBits | ((uint32_t)kLutBits << 27);
        by = 0;
      } else {
        const int base = tm == TW ? 0 : kLutBits;
        uint32_t len;
        bool b = false;
        y = walk_token(rd, t, by, base, &len, &by, &b);
        badv |= b ? 1u : 0u;
        bp += len - (uint32_t)base;
      }
    }
    tm = ntm; tb = ntb;
    const uint32_t extra = __builtin_amdgcn_ubfe((uint32_t)rd.win, y, y >> 5);
    const uint32_t n = y >> 27;
    rd.consume((int)n);
    if (!CLIP || opa - clip_lo < clip_span) {
      const unsigned long long v = (unsigned long long)by << (8u * (opa & 3u));
      const uint32_t a = opa & ~3u;
      // (HIMG_DEC_KO: timing builds of tools/dec_knockout.sh -- they decode wrongly on purpose)
#if !defined(HIMG_DEC_KO) || !(HIMG_DEC_KO & 1)
      lds_or32(a, (uint32_t)v);
#endif
#if !defined(HIMG_DEC_KO) || !(HIMG_DEC_KO & 3)
      lds_or32(a + 4u, (uint32_t)(v >> 32));
#endif
    }
    opa += ((y >> 10) & 511u) + extra;
    bp += n;
  };
  LoopCount lc;
  while (bp < lim) { HIMG_REGION_BEGIN("dec.write"); step(); lc.step(); HIMG_REGION_END("dec.write"); }
  lc.done(0);
  if (tm != TW) step();
  asm volatile("" :: "v"(rd.pre));   // (ReaderT::retire)
  return badv == 0;
}

// chain: [bp, lim) is a stretch of k_row_count_w's chain of groups (lane_off's valid flag
// 3): the groups taken from bp end exactly at lim, one loop without a token-by-token tail.
template <bool CLIP = false>
__device__ __forceinline__ bool lean_write(GReader &rd, const GrpTables &t, uint32_t bp,
                                           uint32_t lim, uint32_t op, uint8_t *lds_out, uint32_t win_span = 0,
                                           bool chain = false) {
  bool bad = false;
  if (bp < lim) {
    rd.init(bp);
    const int limk = (int)lim - kLutBits;
    uint32_t *o32 = reinterpret_cast<uint32_t *>(lds_out);
    // Long codes: two steps, the table being lane state (see lean_count).
    const uint32_t TM = ((1u << kLutBits) - 1u) << 3, TB = lds_addr(t.grp);
    uint32_t tm = TM, tb = TB;
    auto step = [&]() {
      rd.refill();
      const uint2 e = lds_ld64(((((uint32_t)rd.win) << 3) & tm) + tb);
      uint32_t y = e.y, by = e.x, ntm = TM, ntb = TB, adv = 0;
      // (Unconditional: resetting the table state only on the rare paths -- an else-branch on
      // tm != TM -- trades three v_mov for four scalar instructions and a branch per step and
      // measured 12 % slower in the count kernel, 2 % in the row kernel.)
      if (__builtin_expect(y == 0, 0)) {
        if ((by >> 31) && tm == TM) {
          ntm = ((1u << (by & 255u)) - 1u) << 3;
          ntb = TB + (((1u << kLutBits) + ((by >> 8) & 0xffffu)) << 3);
          y = (uint32_t)kLutBits | ((uint32_t)kLutBits << 27);
          by = 0;
        } else {
          const int base = tm == TM ? 0 : kLutBits;
          uint32_t len;
          y = walk_token(rd, t, by, base, &len, &by, &bad);
          adv = len - (uint32_t)base;
        }
      }
      tm = ntm; tb = ntb;
      const uint32_t extra = __builtin_amdgcn_ubfe((uint32_t)rd.win, y, y >> 5);
      const uint32_t n = y >> 27;
      rd.consume((int)n);
      if (!CLIP || op - (kWinGuard - 8u) < win_span - (kWinGuard - 8u)) {
        const unsigned long long v = (unsigned long long)by << (8u * (op & 3u));
        atomicOr(&o32[op >> 2], (uint32_t)v);
        atomicOr(&o32[(op >> 2) + 1], (uint32_t)(v >> 32));
      }
      op += ((y >> 10) & 511u) + extra;
      bp += adv + n;
    };
    if (chain) return lean_write_chain<CLIP>(rd, t, bp, lim, op, lds_out, win_span);   // (uniform per row)
    while ((int)bp <= limk) step();
    if (tm != TM) step();
    while (bp < lim) {
      uint32_t nbits, cnt, by;
      lean_step<true>(rd, t, true, &nbits, &cnt, &by, &bad);
      if (by && (!CLIP || op - kWinGuard < win_span - kWinGuard)) lds_out[op] = (uint8_t)by;
      op += cnt;
      bp += nbits;
    }
    rd.retire();
  }
  return !bad;
}

// The lane in whose range the block completes: exact token-by-token decode with
// the reference's end-of-block checks (huffman_dec.cpp:353-354,361-417).
// Returns false on a stream error; *end_bp = bit position where the block became
// complete (~0u if it did not).
// CLIP: as in lean_write; op and out_size then both count from the window's guard.
template <bool CLIP = false>
__device__ __forceinline__ bool exact_write_body(GReader &rd, const GrpTables &t, uint32_t bp,
                                                 uint32_t lim, uint32_t op, uint32_t out_size,
                                                 uint8_t *lds_out, uint32_t *end_bp, uint32_t win_span) {
  *end_bp = ~0u;
  // Positions are compared as SIGNED numbers: in a window (CLIP) a lane may start in
  // front of it, its position then lies below zero (everything is far below 2^31).
  const auto before = [](uint32_t a, uint32_t b) { return (int32_t)a < (int32_t)b; };
  if (!(bp < lim) || !before(op, out_size)) return true;
  rd.init(bp);
  bool bad = false;
  // Whole groups first, as long as the block stays incomplete behind them: this lane is
  // alone on its path while fifteen wavefronts wait at the barrier behind the write
  // pass -- token by token over its whole range (~50 tokens, each a dependent LDS
  // round trip) it was HALF of that pass.  The group that completes (or overruns) the
  // block is not taken: the reader is put back in front of it and the loop below goes
  // through it token by token with the reference's checks.
  {
    const int limk = (int)lim - kLutBits;
    uint32_t *o32 = reinterpret_cast<uint32_t *>(lds_out);
    while ((int)bp <= limk) {
      const GReader saved = rd;
      uint32_t nbits, cnt, by;
      bool gbad = false;
      lean_step<true>(rd, t, false, &nbits, &cnt, &by, &gbad);
      if (gbad || !before(op + cnt, out_size)) { rd = saved; break; }
      if (!CLIP || op - (kWinGuard - 8u) < win_span - (kWinGuard - 8u)) {
        const unsigned long long v = (unsigned long long)by << (8u * (op & 3u));
        atomicOr(&o32[op >> 2], (uint32_t)v);
        atomicOr(&o32[(op >> 2) + 1], (uint32_t)(v >> 32));
      }
      op += cnt;
      bp += nbits;
    }
  }
  for (;;) {
    uint32_t nbits, cnt, by;
    lean_step<true>(rd, t, true, &nbits, &cnt, &by, &bad);
    if (bad) return false;
    if (before(out_size, op + cnt)) return false;  // a zero run overruns the block
    if (by && (!CLIP || op - kWinGuard < win_span - kWinGuard)) lds_out[op] = (uint8_t)by;
    op += cnt;
    bp += nbits;
    if (!before(op, out_size)) { *end_bp = bp; return true; }
    if (!(bp < lim)) return true;
  }
}

template <bool CLIP = false>
__device__ __forceinline__ bool exact_write(GReader &rd, const GrpTables &t, uint32_t bp,
                                            uint32_t lim, uint32_t op, uint32_t out_size,
                                            uint8_t *lds_out, uint32_t *end_bp, uint32_t win_span = 0) {
  const bool ok = exact_write_body<CLIP>(rd, t, bp, lim, op, out_size, lds_out, end_bp, win_span);
  rd.retire();   // (whichever way the walk ended: see ReaderT::retire)
  return ok;
}

// Write pass to HBM through 32 KiB LDS windows (zero runs are the window's zero
// fill; windows are flushed with 16-byte stores).  Lane t decodes the tokens that
// start in [bp, lim) and places them from output offset op; `exact` marks the lane
// in whose range the block completes.  win holds kWinBytes / 4 + 1 words.
__device__ __forceinline__ void lean_write_windows(GReader &rd, const GrpTables &tb,
                                                   StreamShared *sh, uint32_t bp, uint32_t lim,
                                                   unsigned long long op, bool exact,
                                                   unsigned long long O0, unsigned long long O1,
                                                   uint32_t out_size,
                                                   unsigned long long endbit_base, uint32_t rel0,
                                                   uint32_t *win, uint8_t *gout) {
  const int tid = threadIdx.x;
  bool done = !(bp < lim) || op >= out_size;
  bool bad = false;
  if (!done) rd.init(bp);
  const int limk = (int)lim - kLutBits;
  for (unsigned long long wb = (O0 / kWinBytes) * kWinBytes; wb < O1; wb += kWinBytes) {
    for (int k = tid; k < kWinBytes / 4 + 1; k += kDecThreads) win[k] = 0;
    __syncthreads();
    const unsigned long long we = wb + kWinBytes;
    while (!done && op < we) {
      // The explicit bytes of a group may not straddle the window end.
      const bool single = exact || (int)bp > limk || op + 4 > we;
      uint32_t nbits, cnt, by;
      lean_step<true>(rd, tb, single, &nbits, &cnt, &by, &bad);
      if (exact && (bad || op + cnt > out_size)) { bad = true; break; }
      const uint32_t o = (uint32_t)(op - wb);
      const unsigned long long v = (unsigned long long)by << (8u * (o & 3u));
      atomicOr(&win[o >> 2], (uint32_t)v);
      atomicOr(&win[(o >> 2) + 1], (uint32_t)(v >> 32));
      op += cnt;
      bp += nbits;
      if (exact && op >= out_size) { sh->endbit = endbit_base + (bp - rel0); done = true; }
      else if (!(bp < lim)) done = true;
    }
    if (bad) { sh->err = 1; done = true; }
    __syncthreads();
    // Flush [max(wb, O0), min(we, O1)): 16-byte stores inside, bytes at the edges
    // (another workgroup may own the rest of an edge group).
    const unsigned long long lo = wb > O0 ? wb : O0, hi = we < O1 ? we : O1;
    const uint32_t l = (uint32_t)(lo - wb), h = (uint32_t)(hi - wb);
    const uint32_t la = (l + 15u) & ~15u, ha = h & ~15u;
    const uint8_t *w8 = reinterpret_cast<const uint8_t *>(win);
    if (la <= ha && (((uintptr_t)(gout + wb)) & 15) == 0) {
      for (uint32_t k = l + tid; k < la; k += kDecThreads) gout[wb + k] = w8[k];
      for (uint32_t k = la / 16 + tid; k < ha / 16; k += kDecThreads) {
        uint4 q;
        q.x = win[4 * k]; q.y = win[4 * k + 1]; q.z = win[4 * k + 2]; q.w = win[4 * k + 3];
        *reinterpret_cast<uint4 *>(gout + wb + 16ull * k) = q;
      }
      for (uint32_t k = ha + tid; k < h; k += kDecThreads) gout[wb + k] = w8[k];
    } else {
      for (uint32_t k = l + tid; k < h; k += kDecThreads) gout[wb + k] = w8[k];
    }
    __syncthreads();
  }
  rd.retire();
}

// Write pass straight to PRE-ZEROED global memory (the LRES symbols: 1/64 of the
// frame, L2 resident): lane t decodes the tokens that start in [bp, lim) and stores
// the non-zero literal bytes of every group at their final positions from op; zero
// runs and literal zeros need no store.  No LDS window, no barrier: the kernel that
// calls this keeps only the decode tables in LDS and lasts as long as one lane's 256
// bits.  `exact` marks the lane in whose range the block completes (token by token
// there, with the reference's end-of-block checks, huffman_dec.cpp:353-354,361-417).
__device__ __forceinline__ void lean_write_global(GReader &rd, const GrpTables &tb, StreamShared *sh,
                                                  uint32_t bp, uint32_t lim, unsigned long long op,
                                                  bool exact, uint32_t out_size,
                                                  unsigned long long endbit_base, uint32_t rel0,
                                                  uint8_t *gout, unsigned long long op_end) {
  if (!(bp < lim) || op >= out_size) return;
  rd.init(bp);
  const int limk = (int)lim - kLutBits;
  bool bad = false;
  if (!exact) {
    // Every lane but the one in which the block completes: lean_write's loop (whole groups,
    // a long code as two steps whose table is lane state) with the store below instead of
    // the OR into LDS; the general step further down costs twice the instructions and was
    // k_lres_write's whole time (0.74 -> 0.55 ms per 128 frames, in front of the row kernel).
    const uint32_t TM = ((1u << kLutBits) - 1u) << 3, TB = lds_addr(tb.grp);
    uint32_t tm = TM, tbase = TB;
    uint32_t o = (uint32_t)op;                       // (the output of one stream is below 2^32 symbols)
    const uint32_t o_end = (uint32_t)op_end;
    auto put = [&](uint32_t by) {
      if (by) {
        if (o + 4u <= o_end) {
          asm volatile("global_store_dword %0, %1, %2" :: "v"(o), "v"(by), "s"(gout) : "memory");
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t b = (by >> (8 * j)) & 255u;
            if (b) gout[o + j] = (uint8_t)b;
          }
        }
      }
    };
    auto step = [&]() {
      rd.refill();
      const uint2 e = lds_ld64(((((uint32_t)rd.win) << 3) & tm) + tbase);
      uint32_t y = e.y, by = e.x, ntm = TM, ntb = TB, adv = 0;
      if (__builtin_expect(y == 0, 0)) {
        if ((by >> 31) && tm == TM) {
          ntm = ((1u << (by & 255u)) - 1u) << 3;
          ntb = TB + (((1u << kLutBits) + ((by >> 8) & 0xffffu)) << 3);
          y = (uint32_t)kLutBits | ((uint32_t)kLutBits << 27);
          by = 0;
        } else {
          const int base = tm == TM ? 0 : kLutBits;
          uint32_t len;
          y = walk_token(rd, tb, by, base, &len, &by, &bad);
          adv = len - (uint32_t)base;
        }
      }
      tm = ntm; tbase = ntb;
      const uint32_t extra = __builtin_amdgcn_ubfe((uint32_t)rd.win, y, y >> 5);
      const uint32_t n = y >> 27;
      rd.consume((int)n);
      put(by);
      o += ((y >> 10) & 511u) + extra;
      bp += adv + n;
    };
    while ((int)bp <= limk) step();
    if (tm != TM) step();
    while (bp < lim) {
      uint32_t nbits, cnt, by;
      lean_step<true>(rd, tb, true, &nbits, &cnt, &by, &bad);
      put(by);
      o += cnt;
      bp += nbits;
    }
    rd.retire();
    if (bad) sh->err = 1;
    return;
  }
  for (;;) {
    uint32_t nbits, cnt, by;
    lean_step<true>(rd, tb, exact || (int)bp > limk, &nbits, &cnt, &by, &bad);
    if (exact && (bad || op + cnt > out_size)) { bad = true; break; }
    // A group's (at most 4) explicit bytes lie inside [op, op + cnt); the bytes of the
    // dword beyond them are zeros of its run or belong to this lane's NEXT groups,
    // which are stored later and in order -- so while the dword stays inside the
    // lane's own output range [.., op_end) it goes out as one (unaligned) dword store,
    // and not at all when it is zero.  Near the end of the range: byte by byte.
    if (by) {
      if (op + 4 <= op_end) {
        asm volatile("global_store_dword %0, %1, %2" :: "v"((uint32_t)op), "v"(by), "s"(gout) : "memory");
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t b = (by >> (8 * j)) & 255u;
          if (b) gout[op + j] = (uint8_t)b;
        }
      }
    }
    op += cnt;
    bp += nbits;
    if (exact && op >= out_size) { sh->endbit = endbit_base + (bp - rel0); break; }
    if (!(bp < lim)) break;
  }
  rd.retire();
  if (bad) sh->err = 1;
}

// The kDecThreads sub-sequences of one chunk of `bits` payload bits (`sb` bits each,
// bits <= kDecThreads * sb).  The boundaries lie at k * sb - S: the whole grid is moved
// to the left so that the LAST active lane is left with kTailBits -- that lane ends up
// alone on the token-by-token path that completes the block (exact_write) while every
// other wavefront of the row waits for it, so its range is kept as short as the grid
// allows (lane 0 takes what is cut off; no shift when that would need a 1025th lane).
// Every kernel that walks a chunk derives the same grid from (bits, sb).
constexpr uint32_t kTailBits = 48;
struct SubGrid {
  uint32_t b0, lim;     // this lane's range [b0, lim), window bit positions
  bool active;
  int last_active;
};
__device__ __forceinline__ SubGrid sub_grid(uint32_t rel0, uint32_t bits, uint32_t sb, int tid) {
  SubGrid q;
  const uint32_t la = (bits - 1u) / sb, L = bits - la * sb;
  const uint32_t S = (L > kTailBits && la + 1u < (uint32_t)kDecThreads) ? sb - (L - kTailBits) : 0u;
  const uint32_t rel_end = rel0 + bits;
  q.b0 = tid ? rel0 + (uint32_t)tid * sb - S : rel0;
  q.lim = rel0 + (uint32_t)(tid + 1) * sb - S;
  if (q.lim > rel_end) q.lim = rel_end;
  q.active = q.b0 < rel_end;
  q.last_active = (int)((bits - 1u + S) / sb);
  return q;
}

// k_row_count's record of one lane of a row (see decode_stream's pre_start / pre_off),
// in registers.
struct PreLane {
  uint32_t start, off, nxt, tot, endrel, valid, rounds;
  uint32_t nstart;   // the right neighbour's start (~0u: none, the lane walks to the end of the chunk)
};

// The fixpoint of a chunk that no count kernel has recorded, out of line: the row kernels reach
// it only for rows of several chunks or with an unusable record, and inlined its loops (and the
// 1024-lane scan behind them) were where most of those kernels' scalar registers went -- 47 to
// 118 SGPRs spilled to VGPR lanes, some reloaded inside the write loops.  Everything crosses
// the call by value (a reader passed by reference would live in scratch).
struct ColdFix { uint32_t start, endpos, cnt, rounds; unsigned long long off, tot; long long c_first; };
__device__ __attribute__((noinline)) ColdFix cold_fixpoint(const uint8_t *p, uint32_t stream_size,
                                                           unsigned long long abs_bit, const uint32_t *grp_lds,
                                                           const uint32_t *nd_lds, StreamShared *sh, bool active,
                                                           uint32_t lim, uint32_t start, uint32_t lead_bits) {
  GrpTables tb;
  tb.grp = reinterpret_cast<const uint2 *>(grp_lds); tb.gx = nullptr; tb.gy = nullptr; tb.nd = nd_lds;
  GReader rd;
  const uint32_t rel0 = rd.attach(p, stream_size, abs_bit);
  ColdFix r;
  r.start = start; r.endpos = start; r.cnt = 0; r.rounds = 0; r.c_first = 0;
  lean_fixpoint(rd, tb, sh, rel0, active, lim, &r.start, &r.endpos, &r.cnt, &r.rounds, false, lead_bits, &r.c_first);
  r.off = block_scan_u64(r.cnt, sh->sm64, &r.tot);
  return r;
}

// Decode one whole stream with one workgroup, chunk after chunk (each chunk's
// first token position is exact because the previous chunk has finished).  The
// sub-sequence length is chosen so that the 1024 lanes cover the remaining payload
// (one chunk, normally).
// FUSED: the whole output (out_size bytes) lives in LDS at `lds_out` (pre-zeroed
// by the caller) and the bytes are OR-ed in there; otherwise they go through the
// LDS window `win` to `gout`.  Returns (to every lane) 0 when the stream is
// accepted like UncompressStream accepts it (huffman_dec.cpp:361-417).
// pre_start / pre_off (optional): the fixpoint of the stream's single chunk computed
// beforehand by k_row_count (pre_off[kDecThreads + 2] != 0 says it is usable).
// GLOBAL (with FUSED false): the output goes straight to `gout`, PRE-ZEROED global
// memory, without the LDS window (win may be nullptr) -- the LRES serial fallback.
// PRIMED: the caller has set sh->err / sh->endbit in front of a barrier of its own.
template <bool FUSED, bool GLOBAL = false, bool PRIMED = false>
__device__ __forceinline__ int decode_stream(const uint8_t *p, uint32_t stream_size, uint32_t pay_off,
                             uint32_t pay_len, uint32_t out_size, const GrpTables &tb,
                             StreamShared *sh, uint8_t *lds_out, uint32_t *win, uint8_t *gout,
                             uint32_t *stats, uint32_t max_sub, uint32_t lead_bits,
                             const uint32_t *pre_start = nullptr, const uint32_t *pre_off = nullptr) {
  const int tid = threadIdx.x;
  if (!PRIMED) {
    if (tid == 0) { sh->err = 0; sh->endbit = ~0ull; }
    __syncthreads();
  }

  const unsigned long long P1 = 8ull * pay_len;  // payload end, in bits from pay_off
  unsigned long long cur = 0;                    // exact bit position of the next token
  unsigned long long O0 = 0;                     // symbols produced so far
  uint32_t st_chunks = 0, st_rounds = 0;
  long long c_sync = 0, c_write = 0, c_r1 = 0, c_t0 = clock64();
  bool synced = false;   // (uniform) a barrier stands between the last write pass and here

  while (cur < P1 && O0 < out_size) {
    synced = false;
    GReader rd;
    const uint32_t rel0 = rd.attach(p, stream_size, 8ull * pay_off + cur);
    ++st_chunks;

    const unsigned long long rem = P1 - cur;
    uint32_t sub = (uint32_t)((rem + kDecThreads - 1) / kDecThreads);
    sub = (sub + 31u) & ~31u;
    sub = sub < kMinSubBits ? kMinSubBits : (sub > max_sub ? max_sub : sub);
    const unsigned long long chunk_bits = (unsigned long long)sub * kDecThreads;
    const uint32_t rel_end = rel0 + (uint32_t)(rem < chunk_bits ? rem : chunk_bits);
    const SubGrid sg = sub_grid(rel0, rel_end - rel0, sub, tid);
    const uint32_t my_b0 = sg.b0, lim = sg.lim;
    const bool active = sg.active;
    const int last_active = sg.last_active;

    uint32_t start = active ? my_b0 : rel_end, endpos = start, cnt = 0;
    unsigned long long tot, off;
    // Where the lane's walk ends.  With k_row_count's records: at its right neighbour's
    // start -- the count kernels divide the chunk at token boundaries of their own choice
    // (k_row_count_w: the first GROUP boundary at or past the nominal one), and a lane
    // owns exactly the tokens in front of its neighbour's first.
    uint32_t wlim = lim;
    const uint32_t pre_valid = cur != 0 ? 0u : pre_off ? pre_off[kDecThreads + 2] : 0u;
    const bool pre = pre_valid != 0, chain = pre_valid == 3u;
    if (pre) {
      const uint32_t ns = tid + 1 < kDecThreads ? pre_start[tid + 1] : ~0u;
      wlim = ns < rel_end - rel0 ? rel0 + ns : rel_end;
    }
    if (pre) {
      // One chunk, fixpoint done by k_row_count at twice the occupancy.
      start = rel0 + pre_start[tid];
      off = pre_off[tid];
      const uint32_t nxt_off = pre_off[tid + 1];   // [kDecThreads] holds the total
      cnt = nxt_off - (uint32_t)off;
      tot = pre_off[kDecThreads];
      if (tid == last_active) endpos = rel0 + pre_off[kDecThreads + 1];
      st_rounds += pre_off[kDecThreads + 3];
    } else {
      const ColdFix cf = cold_fixpoint(p, stream_size, 8ull * pay_off + cur, reinterpret_cast<const uint32_t *>(tb.grp),
                                       tb.nd, sh, active, lim, start, lead_bits);
      start = cf.start; endpos = cf.endpos; cnt = cf.cnt; st_rounds += cf.rounds;
      c_r1 += cf.c_first; off = cf.off; tot = cf.tot;
    }
    { const long long t = clock64(); c_sync += t - c_t0; c_t0 = t; }

    const unsigned long long opl = O0 + off;
    // The lane in whose range the block completes takes the exact path.
    const bool inside = opl + cnt < out_size, exact = !inside && opl < out_size;
    if (FUSED) {
      uint32_t end_bp = ~0u;
      if (inside) {
        if (!lean_write(rd, tb, start, wlim, (uint32_t)opl, lds_out, 0u, chain)) sh->err = 1;
      } else if (exact) {
        if (!exact_write(rd, tb, start, wlim, (uint32_t)opl, out_size, lds_out, &end_bp)) sh->err = 1;
      }
      if (end_bp != ~0u) sh->endbit = cur + (end_bp - rel0);
      __syncthreads();
    } else if (GLOBAL) {
      lean_write_global(rd, tb, sh, start, wlim, opl, exact, out_size, cur, rel0, gout, exact ? opl : opl + cnt);
      __syncthreads();
    } else {
      const unsigned long long O1 = (O0 + tot < out_size) ? O0 + tot : out_size;
      lean_write_windows(rd, tb, sh, start, wlim, opl, exact, O0, O1, out_size, cur, rel0, win, gout);
    }
    { const long long t = clock64(); c_write += t - c_t0; c_t0 = t; }

    // ---- advance to the next chunk ----
    if (tid == last_active) sh->nxt[0] = endpos;
    __syncthreads();
    const uint32_t last_end = sh->nxt[0];
    cur += (unsigned long long)(last_end - rel0);
    O0 += tot;
    if (last_end == rel0) break;  // no progress (cannot happen on a valid stream)
    __syncthreads();
  }
  if (!synced) __syncthreads();

  // ---- accept / reject like UncompressStream (huffman_dec.cpp:361-417) ----
  int bad = sh->err;
  if (O0 < out_size) bad = 1;  // ran out of payload before the block was full
  const unsigned long long E = sh->endbit;
  // AtTheEnd (huffman_dec.cpp:140-145): inside the payload's last byte, or exactly at its end.
  if (!bad && !(E <= P1 && E + 8 > P1 && E > 0)) bad = 1;
  if (tid == 0 && stats) {
    stats[0] = st_chunks; stats[1] = st_rounds;
    if (!FUSED) { stats[2] = 0; stats[3] = (uint32_t)(c_r1 >> 4); }   // fused: the caller's slots
    stats[4] = (uint32_t)(c_sync >> 4); stats[5] = (uint32_t)(c_write >> 4);
    stats[6] = pay_len; stats[7] = out_size;
  }
  return bad;
}

// The row kernel's normal case (4096-pixel rows): ONE chunk whose lanes a count kernel has
// recorded, the record in registers (PreLane).  Nothing of the sub-sequence grid is needed -- a
// lane walks from its recorded start to its right neighbour's -- and where the chunk ends is in
// the record as well: one barrier, behind the write pass.  Returns decode_stream's verdict, or
// -1 (to every lane, before anything is written) when the record is not usable: the caller
// then takes decode_stream_fused_cold.  The caller has set sh->err / sh->endbit in front of a
// barrier of its own (decode_stream's PRIMED).
__device__ __forceinline__ int decode_row_recorded(const uint8_t *p, uint32_t stream_size, uint32_t pay_off,
                                                   uint32_t pay_len, uint32_t out_size, const GrpTables &tb,
                                                   StreamShared *sh, uint8_t *lds_out, uint32_t *stats,
                                                   const PreLane &pl) {
  const int tid = threadIdx.x;
  const unsigned long long P1 = 8ull * pay_len;
  if (pl.valid == 0 || P1 == 0 || out_size == 0) return -1;
  const long long c_t0 = clock64();
  GReader rd;
  const uint32_t rel0 = rd.attach(p, stream_size, 8ull * pay_off);
  const uint32_t rel_end = rel0 + (uint32_t)P1;   // (a recorded row is one chunk: k_row_count)
  const uint32_t start = rel0 + pl.start, opl = pl.off, cnt = pl.nxt - pl.off;
  const uint32_t wlim = pl.nstart < (uint32_t)P1 ? rel0 + pl.nstart : rel_end;
  const bool inside = opl + cnt < out_size, exact = !inside && opl < out_size;
  uint32_t end_bp = ~0u;
  if (inside) {
    if (!lean_write(rd, tb, start, wlim, opl, lds_out, 0u, pl.valid == 3u)) sh->err = 1;
  } else if (exact) {
    if (!exact_write(rd, tb, start, wlim, opl, out_size, lds_out, &end_bp)) sh->err = 1;
  }
  if (end_bp != ~0u) sh->endbit = (unsigned long long)(end_bp - rel0);
  __syncthreads();
  const long long c_t1 = clock64();
  // ---- accept / reject like UncompressStream (huffman_dec.cpp:361-417) ----
  int bad = sh->err;
  if (pl.tot < out_size) bad = 1;  // ran out of payload before the block was full
  const unsigned long long E = sh->endbit;
  // AtTheEnd (huffman_dec.cpp:140-145): inside the payload's last byte, or exactly at its end.
  if (!bad && !(E <= P1 && E + 8 > P1 && E > 0)) bad = 1;
  if (tid == 0 && stats) {
    stats[0] = 1; stats[1] = pl.rounds;
    stats[4] = 0; stats[5] = (uint32_t)((c_t1 - c_t0) >> 4);
    stats[6] = pay_len; stats[7] = out_size;
  }
  return bad;
}

// A row without a usable record (several chunks, a count kernel that gave up): the general
// decoder, OUT OF LINE -- inlined beside the fast path above it was most of the row kernel's
// code and of its scalar register pressure.  Everything crosses the call by value.
__device__ __attribute__((noinline)) int decode_stream_fused_cold(const uint8_t *p, uint32_t stream_size,
                                                                  uint32_t pay_off, uint32_t pay_len, uint32_t out_size,
                                                                  const uint32_t *grp_lds, const uint32_t *nd_lds,
                                                                  StreamShared *sh, uint8_t *lds_out, uint32_t *stats,
                                                                  uint32_t max_sub, uint32_t lead_bits) {
  GrpTables tb;
  tb.grp = reinterpret_cast<const uint2 *>(grp_lds); tb.gx = nullptr; tb.gy = nullptr; tb.nd = nd_lds;
  return decode_stream<true, false, true>(p, stream_size, pay_off, pay_len, out_size, tb, sh, lds_out, nullptr, nullptr,
                                          stats, max_sub, lead_bits);
}

// Tree nodes and decode tables of stream `strm` of frame f -> LDS.
__device__ __forceinline__ void load_dec_tables(const DecWs &ws, const DecFrame *df, int f, int strm,
                                                LdsTables *T) {
  const uint32_t *nodes = ws.nodes + ((size_t)f * 2 + strm) * (kMaxNodes + 1);
  const int nn = min(df->s[strm].num_nodes, kMaxNodes + 1);   // (a frame that failed to parse holds anything)
  for (int k = threadIdx.x; k < nn; k += kDecThreads) T->nd[k] = nodes[k];
  const uint4 *gg = reinterpret_cast<const uint4 *>(ws.grp + ((size_t)f * 2 + strm) * (1u << kLutBits));
  for (int k = threadIdx.x; k < (1 << kLutBits) / 2; k += kDecThreads)
    reinterpret_cast<uint4 *>(T->grp)[k] = gg[k];
  const uint4 *gs = reinterpret_cast<const uint4 *>(ws.sub + ((size_t)f * 2 + strm) * kSubEntries);
  for (int k = threadIdx.x; k < kSubEntries / 2; k += kDecThreads)
    reinterpret_cast<uint4 *>(T->grp + (1 << kLutBits))[k] = gs[k];
}

// ---------------------------------------------------------------------------
// k_dec_huff: generic stream decode to the symbol buffers in HBM.  Block 0 of a
// frame is its LRES stream; blocks 1.. are FRES block rows (only launched for
// rows that do not fit the fused kernel's LDS budget).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kDecThreads) void k_dec_huff(Geom g, DecWs ws, const uint8_t *packed,
                                                          size_t in_stride, const uint32_t *sizes,
                                                          int first_block, int lres_fallback_only,
                                                          int use_row_count) {
  __shared__ uint32_t win[kWinBytes / 4 + 1];
  __shared__ LdsTables T;
  __shared__ StreamShared sh;

  const int blk = blockIdx.x + first_block, f = blockIdx.y;
  DecFrame *df = ws.frames + f;
  // One read for the whole workgroup (other workgroups flag the frame concurrently;
  // a per-lane read could let some waves leave ahead of the barriers below).
  if (threadIdx.x == 0) sh.flag = df->status;
  __syncthreads();
  if (sh.flag) return;
  const int strm = blk == 0 ? 0 : 1;
  if (strm == 0 && lres_fallback_only && ws.ver_ok[f]) return;  // parallel LRES path succeeded
  const uint8_t *p = packed + (size_t)f * in_stride;
  uint32_t pay_off, pay_len, out_size;
  uint8_t *out;
  const uint32_t *pre_start = nullptr, *pre_off = nullptr;   // k_row_count's fixpoint, if it ran
  if (strm == 0) {
    pay_off = df->s[0].payload_off;
    pay_len = df->s[0].chunk_end - pay_off;
    out_size = (uint32_t)g.lres_size;
    out = ws.lres_sym + (size_t)f * ws.lres_stride;
  } else {
    const int r = blk - 1;
    pay_off = ws.row_off[(size_t)f * g.rows + r];
    pay_len = ws.row_len[(size_t)f * g.rows + r];
    out_size = (uint32_t)g.row_block;
    out = ws.fres_sym + (size_t)f * ws.fres_stride + (size_t)r * g.row_block;
    if (use_row_count) {
      pre_start = ws.lane_start + ((size_t)f * g.rows + r) * kDecThreads;
      pre_off = ws.lane_off + ((size_t)f * g.rows + r) * (kDecThreads + kRecHdr);
      // use_row_count == 2: rows with a usable fixpoint were written by k_row_window.
      if (use_row_count == 2 && pre_off[kDecThreads + 2] != 0) return;
    }
  }
  load_dec_tables(ws, df, f, strm, &T);
  __syncthreads();
  const GrpTables tb = tables_of(&T);
  const int bad = decode_stream<false>(p, sizes[f], pay_off, pay_len, out_size, tb, &sh, nullptr, win,
                                       out, ws.stats + ((size_t)f * (g.rows + 1) + blk) * 8,
                                       (uint32_t)g.max_sub, (uint32_t)g.lead_bits, pre_start, pre_off);
  if (bad && threadIdx.x == 0) atomicMax(&df->status, fmt_err(strm == 0 ? 4 : 7, 1));
}

// ---------------------------------------------------------------------------
// Parallel LRES decode.  The LRES payload is ONE Huffman stream (~1 M symbols at
// 4096x4096), a long pole for a single workgroup.  All of its chunks are decoded
// speculatively in parallel and the chain between chunks is then made exact:
//   k_lres_spec    chunk k assumes its first token starts at its nominal first
//                  bit and runs the in-chunk fixpoint; stores every lane's start,
//                  end and symbol count, the chunk's end position and total.
//   k_lres_fix     one workgroup per frame: chunk k (k >= 1) is re-based on
//                  T_k = spec end of chunk k-1 -- by one sub-sequence of work when
//                  the chain re-joins the speculative one at once (the normal case),
//                  by a WARM restart of the chunk's fixpoint otherwise -- and the
//                  chain of ends is verified (see the kernel).
//   k_lres_write   chunk k writes its symbols from the corrected lane starts.
// If an end did change (a mis-speculation ran through a whole 32 KiB chunk) the
// frame falls back to the serial workgroup-per-stream path (k_dec_huff).
// ---------------------------------------------------------------------------
constexpr int kLresChunkBits = kDecThreads * kLresSubBits;

// One chunk of the LRES stream, tables already in LDS.  FIX false: the speculative
// in-chunk fixpoint from the nominal first bit.  FIX true: the warm restart with lane
// 0 at T = the speculative end of the previous chunk (the caller has established that
// the cheap test failed).
// The chunk's payload is staged in LDS first (s_pay, kLresPayWords dwords): the rounds
// of a chain that does not self-synchronise -- the run tokens of a constant plane, e.g.
// opaque alpha, repeat with a period and keep a wrong phase for their whole length --
// are ONE lane decoding one sub-sequence each, i.e. pure load latency, and that of the
// LDS is a tenth of the L2's.
constexpr int kLresPayWords = kDecThreads * kLresSubBits / 32 + 8;
constexpr int kLresMemo = kLresMemoWords;   // starts a lane remembers
template <bool FIX, bool STAGE>
__device__ __forceinline__ void lres_chain_body(const Geom &g, const DecWs &ws, const uint8_t *p, uint32_t stream_size,
                                const DecFrame *df, int f, int k, const GrpTables &tb, StreamShared *sh,
                                uint32_t *s_pay) {
  const int tid = threadIdx.x;
  const uint32_t pay_off = df->s[0].payload_off;
  const unsigned long long P1 = 8ull * (df->s[0].chunk_end - pay_off);
  const unsigned long long cur = (unsigned long long)k * kLresChunkBits;
  const size_t slot = (size_t)f * ws.lres_chunks + k;
  uint64_t *end_out = FIX ? ws.fix_end : ws.spec_end;
  GReader gr;
  const uint32_t rel0 = gr.attach(p, stream_size, 8ull * pay_off + cur);
  typename std::conditional<STAGE, LReader, GReader>::type rd;
  if constexpr (STAGE) {
    // Bits up to rel0 + chunk + 46 are consumed and the reader runs three dwords ahead:
    // all inside the staged words, the clamp at jmax is never the stream's data.
    stage_payload(gr, s_pay, (uint32_t)kLresPayWords);
    __syncthreads();
    rd.w = (const __attribute__((address_space(3))) uint32_t *)s_pay;
    rd.jmax = kLresPayWords - 1;
  } else {
    rd = gr;
  }
  const unsigned long long rem = P1 - cur;
  const uint32_t rel_end = rel0 + (uint32_t)(rem < (unsigned long long)kLresChunkBits ? rem : kLresChunkBits);
  const uint32_t my_b0 = rel0 + tid * kLresSubBits;
  uint32_t lim = my_b0 + kLresSubBits;
  if (lim > rel_end) lim = rel_end;
  const bool active = my_b0 < rel_end;
  const int last_active = (int)((rel_end - rel0 - 1u) / kLresSubBits);
  uint32_t start = active ? my_b0 : rel_end, endpos = start, cnt = 0, rounds = 0, first = rel0;
  if (FIX) {
    const unsigned long long T = ws.spec_end[slot - 1];
    // A token is at most 46 bits, so the true first token lies in lane 0's range.
    if (T < cur || T >= cur + kLresSubBits) {
      if (tid == 0) ws.fix_end[slot] = ~0ull;  // forces the serial path
      return;
    }
    first = rel0 + (uint32_t)(T - cur);
    start = rel0 + ws.spec_start[slot * kDecThreads + tid];
    endpos = rel0 + ws.spec_endpos[slot * kDecThreads + tid];
    cnt = ws.spec_cnt[slot * kDecThreads + tid];
  }
  // The lane's memo (lean_fixpoint), handed from the speculative pass to the correction
  // through ws.spec_memo, one packed word per entry: start - nominal (<= 46: the
  // previous lane's last token) | end - lim (<= 46) << 6 | symbols (< 2^20) << 12.
  // Lane 0's start is given, it has no use for one.
  uint32_t memo[3 * kLresMemo];
  uint32_t *gm = ws.spec_memo + (slot * kDecThreads + tid) * kLresMemo;
#pragma unroll
  for (int e = 0; e < kLresMemo; ++e) {
    memo[3 * e] = ~0u; memo[3 * e + 1] = memo[3 * e + 2] = 0;
    if (FIX && tid > 0) {
      const uint32_t m = gm[e];
      if (m != ~0u) { memo[3 * e] = my_b0 + (m & 63u); memo[3 * e + 1] = lim + ((m >> 6) & 63u); memo[3 * e + 2] = m >> 12; }
    }
  }
  const long long c_in = clock64();
  lean_fixpoint<false, decltype(rd), kLresMemo>(rd, tb, sh, first, active, lim, &start, &endpos, &cnt, &rounds, FIX,
                                           (uint32_t)g.lead_bits, nullptr, nullptr, memo);
  const long long c_fix = clock64() - c_in;
  if (!FIX) {
#pragma unroll
    for (int e = 0; e < kLresMemo; ++e) {
      const uint32_t ds = memo[3 * e] - my_b0, de = memo[3 * e + 1] - lim;
      gm[e] = (memo[3 * e] == ~0u || tid == 0 || ds > 46u || de > 46u || memo[3 * e + 2] >= (1u << 20))
                  ? ~0u : (ds | (de << 6) | (memo[3 * e + 2] << 12));
    }
  }
  unsigned long long tot;
  block_scan_u64(cnt, sh->sm64, &tot);
  ws.spec_start[slot * kDecThreads + tid] = start - rel0;
  ws.spec_endpos[slot * kDecThreads + tid] = endpos - rel0;
  ws.spec_cnt[slot * kDecThreads + tid] = cnt;
  if (tid == last_active) { end_out[slot] = cur + (endpos - rel0); ws.spec_tot[slot] = tot; }
  if (tid == 0) {
    uint32_t *st = ws.stats + ((size_t)f * (g.rows + 1)) * 8;
    atomicAdd(&st[FIX ? 2 : 0], 1u);
    atomicAdd(&st[FIX ? 3 : 1], rounds);
    atomicMax(&st[FIX ? 5 : 4], rounds);   // the slowest chunk sets the kernel's duration
    atomicMax(&st[FIX ? 7 : 6], (uint32_t)(c_fix >> 4));
  }
}

// k_lres_spec: every chunk of every frame, speculatively, in parallel.
template <bool STAGE>
__global__ __launch_bounds__(kDecThreads) void k_lres_spec(Geom g, DecWs ws, const uint8_t *packed,
                                                           size_t in_stride, const uint32_t *sizes) {
  __shared__ LdsTables T;
  __shared__ StreamShared sh;
  __shared__ __attribute__((aligned(16))) uint32_t s_pay[STAGE ? kLresPayWords : 4];
  const int k = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
  DecFrame *df = ws.frames + f;
  if (tid == 0) sh.flag = df->status;
  __syncthreads();
  if (sh.flag) return;
  const unsigned long long P1 = 8ull * (df->s[0].chunk_end - df->s[0].payload_off);
  const unsigned long long cur = (unsigned long long)k * kLresChunkBits;
  const size_t slot = (size_t)f * ws.lres_chunks + k;
  if (cur >= P1) {
    if (tid == 0) { ws.spec_end[slot] = cur; ws.spec_tot[slot] = 0; }
    return;
  }
  load_dec_tables(ws, df, f, 0, &T);
  __syncthreads();
  const GrpTables tb = tables_of(&T);
  lres_chain_body<false, STAGE>(g, ws, packed + (size_t)f * in_stride, sizes[f], df, f, k, tb, &sh, s_pay);
}

// k_lres_fix: correction and verification of ALL chunks of a frame by one workgroup.
//   1. lane k = chunk k.  Chunk k's true first token starts where chunk k-1's
//      speculative chain ended (T); lane k decodes the first sub-sequence of its chunk
//      from T, and if that ends where the speculative first lane ended, nothing else in
//      the chunk changes: the chunk is settled with one sub-sequence of work (this
//      used to cost a whole workgroup and a load of the decode tables per chunk);
//   2. the chunks that did not re-join (none, normally) take the warm restart, one
//      after the other, with the whole workgroup;
//   3. chunk 0 is exact, so T_1 and hence chunk 1's corrected chain are exact; T_2 was
//      taken from chunk 1's SPECULATIVE end, which is right iff the correction left
//      that end unchanged -- and so on.  All ends unchanged => every chunk exact
//      (induction).  Output offsets = scan of the corrected totals.
template <bool STAGE>
__global__ __launch_bounds__(kDecThreads) void k_lres_fix(Geom g, DecWs ws, const uint8_t *packed,
                                                          size_t in_stride, const uint32_t *sizes) {
  __shared__ LdsTables T;
  __shared__ StreamShared sh;
  __shared__ uint8_t s_pending[kDecThreads];
  __shared__ int s_bad;
  __shared__ __attribute__((aligned(16))) uint32_t s_pay[STAGE ? kLresPayWords : 4];
  const int f = blockIdx.x, k = threadIdx.x;
  DecFrame *df = ws.frames + f;
  if (k == 0) { ws.ver_ok[f] = 0; ws.lres_endbit[f] = ~0ull; s_bad = 0; sh.flag = df->status; }
  s_pending[k] = 0;
  __syncthreads();
  if (sh.flag) return;
  const uint32_t pay_off = df->s[0].payload_off;
  const unsigned long long P1 = 8ull * (df->s[0].chunk_end - pay_off);
  const int nact = (int)((P1 + kLresChunkBits - 1) / kLresChunkBits);
  if (nact > ws.lres_chunks || nact > kDecThreads) return;  // ver_ok stays 0 -> serial path
  load_dec_tables(ws, df, f, 0, &T);
  __syncthreads();
  const GrpTables tb = tables_of(&T);
  const uint8_t *p = packed + (size_t)f * in_stride;
  const size_t slot = (size_t)f * ws.lres_chunks + k;
  const unsigned long long cur = (unsigned long long)k * kLresChunkBits;
  // ---- 1: the cheap test, all chunks at once
  if (k == 0) {
    ws.fix_end[slot] = ws.spec_end[slot];   // chunk 0 is exact already
  } else if (k < nact) {
    const unsigned long long T = ws.spec_end[slot - 1];
    bool settled = false;
    if (T >= cur && T < cur + kLresSubBits) {
      GReader rd;
      const uint32_t rel0 = rd.attach(p, sizes[f], 8ull * pay_off + cur);
      const unsigned long long rem = P1 - cur;
      uint32_t lim = rel0 + kLresSubBits;
      const uint32_t rel_end = rel0 + (uint32_t)(rem < (unsigned long long)kLresChunkBits ? rem : kLresChunkBits);
      if (lim > rel_end) lim = rel_end;
      uint32_t pos, c;
      lean_count(rd, tb, rel0 + (uint32_t)(T - cur), lim, &pos, &c);
      if (pos - rel0 == ws.spec_endpos[slot * kDecThreads]) {
        const unsigned long long oldc = ws.spec_cnt[slot * kDecThreads];
        ws.spec_start[slot * kDecThreads] = (uint32_t)(T - cur);
        ws.spec_cnt[slot * kDecThreads] = c;
        ws.spec_tot[slot] = ws.spec_tot[slot] - oldc + c;
        ws.fix_end[slot] = ws.spec_end[slot];
        settled = true;
        atomicAdd(&ws.stats[((size_t)f * (g.rows + 1)) * 8 + 2], 1u);
      }
    }
    if (!settled) s_pending[k] = 1;
  }
  __syncthreads();
  // ---- 2: warm restarts of what is left
  for (int q = 1; q < nact; ++q) {
    if (!s_pending[q]) continue;   // LDS: the same for every lane
    lres_chain_body<true, STAGE>(g, ws, p, sizes[f], df, f, q, tb, &sh, s_pay);
    __syncthreads();
  }
  // ---- 3: verification and output offsets (other lanes' global writes of this
  // kernel are read below)
  __threadfence_block();
  __syncthreads();
  unsigned long long ntot = 0;
  if (k < nact) {
    ntot = ws.spec_tot[slot];
    // The next chunk was corrected against spec_end[k]; that is the true end of
    // chunk k iff the correction of chunk k did not move it.
    if (k + 1 < nact && ws.fix_end[slot] != ws.spec_end[slot]) s_bad = 1;
  }
  unsigned long long tot;
  const unsigned long long base = block_scan_u64(ntot, sh.sm64, &tot);
  if (k < nact) ws.ver_base[slot] = base;
  __syncthreads();
  if (k == 0) ws.ver_ok[f] = (s_bad || g.lres_serial) ? 0 : 1;
}

__global__ __launch_bounds__(kDecThreads) void k_lres_write(Geom g, DecWs ws, const uint8_t *packed,
                                                            size_t in_stride, const uint32_t *sizes) {
  __shared__ LdsTables T;
  __shared__ StreamShared sh;
  const int k = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
  DecFrame *df = ws.frames + f;
  // Everything the common path reads from global memory is requested in front of the
  // first barrier (the verdict, the chunk's output base, the lane's start and count,
  // the tables), so that the round trips overlap: the kernel is a handful of dependent
  // round trips long and runs on an empty GPU in front of the row kernel.
  const size_t slot = (size_t)f * ws.lres_chunks + k;
  const unsigned long long O0 = ws.ver_base[slot];
  const uint32_t start_rel = ws.spec_start[slot * kDecThreads + tid];
  const unsigned long long cnt = ws.spec_cnt[slot * kDecThreads + tid];
  // One read for the whole workgroup (other kernels may flag the frame meanwhile).
  if (tid == 0) sh.flag = df->status ? 1 : (ws.ver_ok[f] ? 0 : 2);
  load_dec_tables(ws, df, f, 0, &T);
  __syncthreads();
  if (sh.flag == 1) return;
  if (sh.flag == 2) {
    // The chunk chain did not verify (a mis-speculation ran through a whole chunk, or
    // the test knob): this frame's LRES stream is decoded by ONE workgroup, chunk
    // after chunk, every chunk starting at the exact end of the one before.
    if (k != 0) return;
    const GrpTables tb0 = tables_of(&T);
    const uint32_t po = df->s[0].payload_off;
    const int bad = decode_stream<false, true>(packed + (size_t)f * in_stride, sizes[f], po, df->s[0].chunk_end - po,
                                               (uint32_t)g.lres_size, tb0, &sh, nullptr, nullptr,
                                               ws.lres_sym + (size_t)f * ws.lres_stride,
                                               ws.stats + ((size_t)f * (g.rows + 1)) * 8, (uint32_t)g.max_sub,
                                               (uint32_t)g.lead_bits);
    if (bad && tid == 0) atomicMax(&df->status, fmt_err(4, 1));
    return;
  }
  const uint32_t pay_off = df->s[0].payload_off;
  const unsigned long long P1 = 8ull * (df->s[0].chunk_end - pay_off);
  const unsigned long long cur = (unsigned long long)k * kLresChunkBits;
  if (cur >= P1) return;
  const uint32_t out_size = (uint32_t)g.lres_size;
  if (O0 >= out_size) return;
  if (tid == 0) { sh.err = 0; sh.endbit = ~0ull; }
  const GrpTables tb = tables_of(&T);
  GReader rd;
  const uint32_t rel0 = rd.attach(packed + (size_t)f * in_stride, sizes[f], 8ull * pay_off + cur);
  const unsigned long long rem = P1 - cur;
  const uint32_t rel_end = rel0 + (uint32_t)(rem < (unsigned long long)kLresChunkBits ? rem : kLresChunkBits);
  uint32_t lim = rel0 + (tid + 1) * kLresSubBits;
  if (lim > rel_end) lim = rel_end;
  const uint32_t start = rel0 + start_rel;
  __syncthreads();
  unsigned long long tot;
  const unsigned long long off = block_scan_u64(cnt, sh.sm64, &tot);
  const unsigned long long opl = O0 + off;
  const bool exact = !(opl + cnt < out_size) && opl < out_size;
  // The symbols were zeroed by launch_decode: only non-zero literals are stored.
  lean_write_global(rd, tb, &sh, start, lim, opl, exact, out_size, cur, rel0,
                    ws.lres_sym + (size_t)f * ws.lres_stride, exact ? opl : opl + cnt);
  __syncthreads();
  if (tid == 0) {
    if (sh.err) atomicMax(&df->status, fmt_err(4, 1));
    if (sh.endbit != ~0ull) ws.lres_endbit[f] = sh.endbit;
  }
}

// Accept / reject the LRES stream like UncompressStream (huffman_dec.cpp:361-417).
__global__ void k_lres_finish(DecWs ws, int batch) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= batch) return;
  DecFrame *df = ws.frames + f;
  if (df->status || !ws.ver_ok[f]) return;  // failed earlier, or handled by the serial path
  const unsigned long long P1 = 8ull * (df->s[0].chunk_end - df->s[0].payload_off);
  const unsigned long long E = ws.lres_endbit[f];
  if (!(E != ~0ull && E <= P1 && E + 8 > P1 && E > 0)) atomicMax(&df->status, fmt_err(4, 1));
}

// ---------------------------------------------------------------------------
// k_lres_unpredict: one wavefront per 16x16 macro block, anti-diagonal
// wavefronts like k_lres_predict (downsampled.cpp:318-382).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int predict_d(int s1, int s2, int s3, int p) {
  switch (p) {
    default:
    case 0: return clamp255d((3 * (s2 + s3) - 2 * s1 + 2) >> 2);
    case 1: return s2;
    case 2: return s3;
    case 3: return (s2 + s3 + 1) >> 1;
    case 4: return clamp255d(s2 + s3 - s1);
  }
}

// LDS exchange inside ONE wavefront: its LDS operations execute in order, so the
// fences only keep the compiler from moving them.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sixteen macro blocks per workgroup: four wavefronts, each on its own (four blocks of
// 16 lanes, see k_lres_predict) -- one-wavefront workgroups made this kernel a test of
// the dispatcher (65 536 workgroups for 64 frames of 4096 x 4096), and it sits on the
// critical path between k_row_count and the row kernel.
constexpr int kUnpredWaves = 4;
__global__ __launch_bounds__(64 * kUnpredWaves) void k_lres_unpredict(Geom g, DecWs ws) {
  __shared__ uint8_t rec_s[kUnpredWaves][4][17][18];   // the reconstructed blocks with a border row / column in front
  __shared__ uint8_t dl_s[kUnpredWaves][4][16][16];   // the blocks' deltas: the chain below reads one per step
  __shared__ int16_t s_lmap[128];   // the chain below looks a delta up per step: LDS, not global
  __shared__ int s_status;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = lane >> 4, dv = lane & 15;
  uint8_t (*rec)[17][18] = rec_s[wv];
  uint8_t (*dl)[16][16] = dl_s[wv];
  const int mu = (blockIdx.x * kUnpredWaves + wv) * 4 + b, mv = blockIdx.y;
  const int f = blockIdx.z / g.C, c = blockIdx.z % g.C;
  const DecFrame *df = ws.frames + f;
  if (threadIdx.x == 0) s_status = df->status;
  if (threadIdx.x < 128) s_lmap[threadIdx.x] = df->lmap[threadIdx.x];
  const uint8_t *in = ws.lres_sym + (size_t)f * ws.lres_stride + (size_t)c * g.chan_size;
  uint8_t *m = ws.low + (size_t)f * ws.plane_stride + (size_t)c * g.rows * g.cols;
  const bool live = mu < g.mcols;
  const int u0 = mu * 16, v0 = mv * 16;
  const int bw = live ? min(16, g.cols - u0) : 0, bh = min(16, g.rows - v0);
  // DecodePredictor (downsampled.cpp:37-39): uint8 + 2 in int arithmetic, so
  // stored 254/255 come back as 256/257 and fall into PredictSample's default.
  const int pc = live ? (int)in[mv * g.mcols + mu] + 2 : 0;
  const uint8_t *src = in + g.mrows * g.mcols + (size_t)v0 * g.cols + (size_t)bh * u0;
  // The lane's row of deltas goes to LDS up front (sixteen loads in flight at once)
  // instead of one dependent global load per step of the chain.
  if (dv < bh) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < bw) dl[b][dv][k] = src[dv * bw + k];
  }
  __syncthreads();
  if (s_status) return;
  // The chain, branch free (as k_lres_predict's): neighbours from a copy of the block with a border
  // (index + 1), the edge cases of downsampled.cpp:318-382 as four selects, all predictors computed
  // and the block's selected -- the four blocks of a wavefront use different ones.
  const bool row_live = dv < bh, up_ok = dv > 0;
  uint8_t *rrow = &rec[b][dv + 1][1];          // rrow[du] = reconstructed sample (dv, du)
  const uint8_t *urow = &rec[b][dv][1];        // the row above
  uint8_t *mrow = m + (size_t)(v0 + dv) * g.cols + u0;
  for (int d = 0; d < 31; ++d) {
    const int du = d - dv;
    const bool active = row_live && du >= 0 && du < bw;
    const int duc = active ? du : 0;
    const int left = rrow[duc - 1], up = urow[duc], ul = urow[duc - 1];
    const bool left_ok = duc > 0;
    const int f = up_ok ? up : (left_ok ? left : 128);
    const int s3 = left_ok ? left : f, s2 = f, s1 = (up_ok && left_ok) ? ul : f;
    const int t = s2 + s3;
    const int p0 = clamp255d((3 * t - 2 * s1 + 2) >> 2), p3 = (t + 1) >> 1, p4 = clamp255d(t - s1);
    const int predicted = pc == 1 ? s2 : pc == 2 ? s3 : pc == 3 ? p3 : pc == 4 ? p4 : p0;
    const int sc = (int8_t)dl[b][dv][duc];
    // mapper.h:33-35 with the mirrored table (mapper.cpp:148-154).
    const int mag = s_lmap[sc < 0 ? (sc == -128 ? 127 : -sc) : sc];
    const int un = sc < 0 ? -mag : mag;
    const int val = clamp255d((int)(int16_t)(predicted + un));
    if (active) {
      rrow[du] = (uint8_t)val;
      mrow[du] = (uint8_t)val;
    }
    wave_lds_sync();   // (the exchange is inside the wavefront)
  }
}

// ---------------------------------------------------------------------------
// Inverse transform (shared by k_dec_row_fused and k_tile_inv).
// ---------------------------------------------------------------------------
// Inverse 8-point butterfly: int32, floor >>3, narrowed to int16 (hadamard.cpp:47-74).
__device__ __forceinline__ void iwht8(int &x0, int &x1, int &x2, int &x3, int &x4, int &x5,
                                      int &x6, int &x7) {
  const int a0 = x0 + x4, a1 = x1 + x5, a2 = x2 + x6, a3 = x3 + x7;
  const int a4 = x0 - x4, a5 = x1 - x5, a6 = x2 - x6, a7 = x3 - x7;
  const int b0 = a0 + a2, b1 = a1 + a3, b2 = a0 - a2, b3 = a1 - a3;
  const int b4 = a4 + a6, b5 = a5 + a7, b6 = a4 - a6, b7 = a5 - a7;
  x0 = (int16_t)((b0 + b1) >> 3); x1 = (int16_t)((b4 + b5) >> 3);
  x2 = (int16_t)((b6 + b7) >> 3); x3 = (int16_t)((b2 + b3) >> 3);
  x4 = (int16_t)((b2 - b3) >> 3); x5 = (int16_t)((b6 - b7) >> 3);
  x6 = (int16_t)((b4 - b5) >> 3); x7 = (int16_t)((b0 - b1) >> 3);
}

// ---------------------------------------------------------------------------
// k_dec_row_fused: one 1024-lane workgroup per FRES block row, when the row's
// symbols fit in LDS (row_block <= ~128 KiB, i.e. width <= 4096 for RGBA).
//   1. entropy-decode the row into LDS (decode_stream<true>);
//   2. lane = tile: gather, dequantise, inverse WHT, + low-res, clamp; results
//      overwrite the tile's own 64 symbol slots per channel (in place);
//   3. lane = (tile, pixel row): colour inverse and two 16-byte stores, so a wave
//      writes 2 KiB of contiguous pixels per pixel row.
// The symbols never touch HBM: traffic is the packed row in, the pixels out.
// ---------------------------------------------------------------------------
struct FusedLayout {
  uint32_t sym, tab, sh, unmap, shift, shiftp, total;
};
__host__ __device__ inline FusedLayout fused_layout(int row_block, int rows = 1) {
  FusedLayout L;
  uint32_t o = 0;
  auto carve = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15u) & ~15u; return r; };
  L.tab = carve((uint32_t)sizeof(LdsTables));   // at offset 0: the hot loop indexes it
  L.sym = carve((((uint32_t)row_block + 15u) & ~15u) * (uint32_t)rows);   // `rows` block rows of symbols
  L.sh = carve((uint32_t)sizeof(StreamShared));
  L.unmap = carve(512u);
  L.shift = carve(128u);
  L.shiftp = carve(256u + 16u);   // 2 x 32 packed shift pairs, then kHfast words (tile_plane's identity test)
  L.total = o;
  return L;
}

// ---------------------------------------------------------------------------
// Packed-int16 inverse transform of one 8x8 plane (one channel of one tile).
//
// The reference does every butterfly in int32 and narrows after the >>3
// (hadamard.cpp:47-74), so 16-bit lanes are only exact while no sum of eight
// terms leaves int16.  With every dequantised coefficient in [-4096, 4095] that
// holds for both passes (|sum| <= 32767 + the one -32768 that still fits, and a
// pass maps the range onto itself), which is the case for anything but
// synthetic full-swing noise; a plane with a larger coefficient takes the
// scalar int32 path below.  Both paths are bit-exact.
//
// Layout of the packed path: V[x*4 + j] = rows (2j, 2j+1) of column x, so the
// row pass (along x) is element-wise over registers; one v_perm per register
// pair turns that into T[y*4 + i] = columns (2i, 2i+1) of row y for the column
// pass (along y).  The bilinear low-res block (downsampled.cpp:116-169) is
// built four rows per register with v_lerp_u8 ((a + b + 1) >> 1 per byte), the
// int16 add + ClampTo8Bit (decoder.cpp:401-413) is pk_add + v_sat_pk_u8_i16.
// Output: O[y*2 + h] = the clamped bytes of row y, x = 4h..4h+3.
// ---------------------------------------------------------------------------
typedef short dpk16 __attribute__((ext_vector_type(2)));
typedef unsigned short dupk16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sat_pk_u8(uint32_t x) {
  uint32_t r;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(x));
  return r;   // byte 0 = sat(lo half), byte 1 = sat(hi half)
}

// The same into the HIGH half of `lo` (whose low half holds an earlier pair's two bytes): SDWA
// writes the result word there and keeps the other, which saves the v_perm that would merge
// two results.
__device__ __forceinline__ uint32_t sat_pk_u8_hi(uint32_t lo, uint32_t x) {
  asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(lo) : "v"(x));
  return lo;
}

// 2 * a + c on packed int16 pairs, one instruction.
__device__ __forceinline__ uint32_t pk_mad2(dpk16 a, dpk16 c) {
  uint32_t r;
  asm("v_pk_mad_i16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, a)), "v"(__builtin_bit_cast(uint32_t, c)));
  return r;
}

__device__ __forceinline__ void iwht8_pk(dpk16 &x0, dpk16 &x1, dpk16 &x2, dpk16 &x3, dpk16 &x4,
                                         dpk16 &x5, dpk16 &x6, dpk16 &x7) {
  const dpk16 three = {3, 3};
  const dpk16 a0 = x0 + x4, a1 = x1 + x5, a2 = x2 + x6, a3 = x3 + x7;
  const dpk16 a4 = x0 - x4, a5 = x1 - x5, a6 = x2 - x6, a7 = x3 - x7;
  const dpk16 b0 = a0 + a2, b1 = a1 + a3, b2 = a0 - a2, b3 = a1 - a3;
  const dpk16 b4 = a4 + a6, b5 = a5 + a7, b6 = a4 - a6, b7 = a5 - a7;
  x0 = (b0 + b1) >> three; x1 = (b4 + b5) >> three;
  x2 = (b6 + b7) >> three; x3 = (b2 + b3) >> three;
  x4 = (b2 - b3) >> three; x5 = (b6 - b7) >> three;
  x6 = (b4 - b5) >> three; x7 = (b0 - b1) >> three;
}

// lerp tree of downsampled.cpp:116-169 on four packed bytes.
__device__ __forceinline__ void interp9_u8x4(uint32_t a[9]) {
  const uint32_t rnd = 0x01010101u;
  a[4] = __builtin_amdgcn_lerp(a[0], a[8], rnd);
  a[2] = __builtin_amdgcn_lerp(a[0], a[4], rnd);
  a[6] = __builtin_amdgcn_lerp(a[4], a[8], rnd);
  a[1] = __builtin_amdgcn_lerp(a[0], a[2], rnd);
  a[3] = __builtin_amdgcn_lerp(a[2], a[4], rnd);
  a[5] = __builtin_amdgcn_lerp(a[4], a[6], rnd);
  a[7] = __builtin_amdgcn_lerp(a[6], a[8], rnd);
}

// Low-res block: LQ[q][x] = bytes of rows 4q..4q+3 at column x.
__device__ __forceinline__ void lowres_quads(uint32_t lr0, uint32_t lr8, uint32_t LQ[2][8]) {
  uint32_t lr[9];
  lr[0] = lr0; lr[8] = lr8;
  interp9_u8x4(lr);   // bytes 0,1 = left,right of rows 0..7
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t p01 = __builtin_amdgcn_perm(lr[4 * q + 1], lr[4 * q], 0x05010400u);      // L0 L1 R0 R1
    const uint32_t p23 = __builtin_amdgcn_perm(lr[4 * q + 3], lr[4 * q + 2], 0x05010400u);  // L2 L3 R2 R3
    uint32_t a[9];
    a[0] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
    a[8] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
    interp9_u8x4(a);
#pragma unroll
    for (int x = 0; x < 8; ++x) LQ[q][x] = a[x];
  }
}

// When are 16-bit lanes exact for the two inverse passes?  The reference adds in int32
// and narrows after each pass's >> 3 (hadamard.cpp:47-74), so all that is needed is
// that no butterfly sum leaves int16, i.e. sum |d| <= 32767 over every ROW going into
// the row pass and over every column of its results.  Two sufficient conditions on the
// dequantised coefficients, checked from running maxima / minima (v_pk_max_i16 /
// v_pk_min_i16):
//   (A) |d| <= 3071 everywhere but in register 0 (rows 0 and 1 of column 0: the DC and
//       its vertical neighbour), where |d| <= 11263:  rows 0, 1: 11263 + 7 x 3071 =
//       32760; other rows 8 x 3071; after >> 3 at most 4095 / 3071, so a column sums to
//       at most 2 x 4095 + 6 x 3071 = 26616;
//   (B) |d| <= 4095 everywhere (8 x 4095 = 32760 in both passes).
// (B) alone -- the round-2 test -- fails in nine of ten wavefronts of the benchmark
// frames, whose DC terms reach 8320 (alpha: 10448): a tile whose neighbours differ needs
// a large DC correction on top of the bilinear low-res plane, and ONE such lane sent its
// whole wavefront through the scalar int32 path as well.  (A) holds for every tile of
// those frames, (B) covers full-swing noise with small DC terms; a plane that satisfies
// neither takes the scalar path, which is exact for anything.
struct RangeAcc {
  dpk16 mx, mn;
  __device__ __forceinline__ void init() { mx = dpk16{-32768, -32768}; mn = dpk16{32767, 32767}; }
  __device__ __forceinline__ void add(dpk16 d) { mx = __builtin_elementwise_max(mx, d); mn = __builtin_elementwise_min(mn, d); }
};
__device__ __forceinline__ bool packed_wht_exact(const RangeAcc &others, dpk16 d0) {
  const dpk16 zero = {0, 0};
  const dupk16 am = __builtin_bit_cast(dupk16, (dpk16)__builtin_elementwise_max(others.mx, __builtin_elementwise_sub_sat(zero, others.mn)));
  const dupk16 a0 = __builtin_bit_cast(dupk16, (dpk16)__builtin_elementwise_max(d0, __builtin_elementwise_sub_sat(zero, d0)));
  const dupk16 t3071 = {3071, 3071}, t4095 = {4095, 4095}, t11263 = {11263, 11263};
  const uint32_t oa = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(am, t3071)) |
                      __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(a0, t11263));
  const uint32_t ob = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(am, t4095)) |
                      __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(a0, t4095));
  return oa == 0u || ob == 0u;
}

// slot: the tile's first symbol (scan index k is at slot[k * cols]); shift: the
// 64 per-position shifts; shiftp: the same as 32 packed pairs in V order; lr0 /
// lr8: (left, right) low-res samples of this and the next block row in bytes 0, 1.
// hfast (COLS != 0): {bias pair, mask pair} of the identity test below.
// COLS > 0: the tile count per block row at compile time (slot offsets are immediates);
// COLS < 0: the same fast gather with the run-time count (64 address adds per plane).
template <int COLS>
__device__ __forceinline__ void tile_plane(const uint8_t *slot, int cols_rt, const int16_t *s_unmap,
                                           const uint8_t *shift, const uint32_t *shiftp,
                                           uint32_t lr0, uint32_t lr8, uint32_t O[16],
                                           const uint32_t *hfast = nullptr) {
  const int cols = COLS > 0 ? COLS : cols_rt;
  // Gather + dequantise (quantize.cpp:153-165: int16 wrap == 16-bit shift left).
  dpk16 V[32];
  RangeAcc rng;
  rng.init();
  if (COLS != 0) {
    // The companding table is the identity for small codes (mapper.cpp:54-61: up to
    // 49 in the encoder's table; the decoder derives the range from the stream's
    // FMAP), and everything but the lowest frequencies of a tile IS small.  The 21
    // register pairs that hold rows 2..7 of columns 1..7 (group H) are therefore
    // loaded as sign-extended bytes straight into the packed halves
    // (ds_read_i8_d16 / _d16_hi: no look-up, no address arithmetic, no v_perm) and
    // tested ONCE per plane and wavefront: every code in [-B, B) with B a power of
    // two inside the identity range and B << (largest H shift) <= 2048, so the test
    // also stands for the 16-bit range check of these coefficients.  A wavefront
    // with a larger code in H takes the look-ups for H as well.  Group L (rows 0, 1
    // and column 0: the DC and first-order coefficients, large in most tiles)
    // always goes through the table.
    const uint32_t hb = hfast[0], hm = hfast[1];
    uint32_t W[32], hacc = 0;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int x = e >> 2, j = e & 3, p0 = (2 * j) * 8 + x;
      const uint8_t *q0 = slot + (size_t)kInvScanD[p0] * cols, *q1 = slot + (size_t)kInvScanD[p0 + 8] * cols;
      if (x == 0 || j == 0) {
        W[e] = (uint32_t)(uint16_t)s_unmap[*q0] | ((uint32_t)(uint16_t)s_unmap[*q1] << 16);
      } else {
        dpk16 w;
        w.x = (short)(int8_t)*q0;
        w.y = (short)(int8_t)*q1;
        W[e] = __builtin_bit_cast(uint32_t, w);
        hacc |= __builtin_bit_cast(uint32_t, (dupk16)(__builtin_bit_cast(dupk16, w) + __builtin_bit_cast(dupk16, hb)));
      }
    }
    const bool hslow = __any((hacc & hm) != 0u);
    if (__builtin_expect(hslow, 0)) {
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        const int x = e >> 2, j = e & 3;
        if (x == 0 || j == 0) continue;
        W[e] = (uint32_t)(uint16_t)s_unmap[W[e] & 255u] | ((uint32_t)(uint16_t)s_unmap[(W[e] >> 16) & 255u] << 16);
      }
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int x = e >> 2, j = e & 3;
      const dupk16 d = __builtin_bit_cast(dupk16, W[e]) << __builtin_bit_cast(dupk16, shiftp[e]);
      V[e] = __builtin_bit_cast(dpk16, d);
      if ((x == 0 || j == 0) && e != 0) rng.add(V[e]);   // (group H: below 2048 by its own test)
    }
    if (__builtin_expect(hslow, 0)) {
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        const int x = e >> 2, j = e & 3;
        if (x == 0 || j == 0) continue;
        rng.add(V[e]);
      }
    }
  } else {
    uint32_t W[32];
    {
      // Run-time column count: walk the slots in scan order with one running
      // pointer (64 separate offsets would not stay in registers).
#pragma unroll
      for (int e = 0; e < 32; ++e) W[e] = 0;
      const uint8_t *q = slot;
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        const int pos = kScanD[k], y = pos >> 3, x = pos & 7;
        const int code = *q;
        q += cols;
        if ((k & 7) == 7) asm volatile("" : "+v"(q));
        W[x * 4 + (y >> 1)] |= (uint32_t)(uint16_t)s_unmap[code] << ((y & 1) * 16);
      }
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const dupk16 d = __builtin_bit_cast(dupk16, W[e]) << __builtin_bit_cast(dupk16, shiftp[e]);
      V[e] = __builtin_bit_cast(dpk16, d);
      if (e != 0) rng.add(V[e]);
    }
  }
  dpk16 T[32];   // T[y*4 + i] = columns (2i, 2i+1) of row y after both passes
  if (__builtin_expect(packed_wht_exact(rng, V[0]), 1)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      iwht8_pk(V[j], V[4 + j], V[8 + j], V[12 + j], V[16 + j], V[20 + j], V[24 + j], V[28 + j]);
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        T[y * 4 + i] = __builtin_bit_cast(
            dpk16, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, V[(2 * i + 1) * 4 + (y >> 1)]),
                                         __builtin_bit_cast(uint32_t, V[(2 * i) * 4 + (y >> 1)]),
                                         (y & 1) ? 0x07060302u : 0x05040100u));
#pragma unroll
    for (int i = 0; i < 4; ++i)
      iwht8_pk(T[i], T[4 + i], T[8 + i], T[12 + i], T[16 + i], T[20 + i], T[24 + i], T[28 + i]);
  } else {
    // Scalar int32 path, one row / two columns at a time.
    uint32_t P[32];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      int r[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const uint32_t w = __builtin_bit_cast(uint32_t, V[x * 4 + (y >> 1)]);
        r[x] = (int)(int16_t)(uint16_t)(w >> ((y & 1) * 16));
      }
      iwht8(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
#pragma unroll
      for (int x = 0; x < 8; x += 2)
        P[y * 4 + x / 2] = ((uint32_t)r[x] & 0xffffu) | ((uint32_t)r[x + 1] << 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ca[8], cb[8];
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        ca[y] = (int)(int16_t)(uint16_t)(P[y * 4 + i] & 0xffffu);
        cb[y] = (int)(int16_t)(uint16_t)(P[y * 4 + i] >> 16);
      }
      iwht8(ca[0], ca[1], ca[2], ca[3], ca[4], ca[5], ca[6], ca[7]);
      iwht8(cb[0], cb[1], cb[2], cb[3], cb[4], cb[5], cb[6], cb[7]);
#pragma unroll
      for (int y = 0; y < 8; ++y)
        T[y * 4 + i] = __builtin_bit_cast(dpk16, ((uint32_t)ca[y] & 0xffffu) | ((uint32_t)cb[y] << 16));
    }
  }
  uint32_t LQ[2][8];
  lowres_quads(lr0, lr8, LQ);
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    uint32_t sum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t sel = 0x0c000c00u | (uint32_t)(y & 3) | ((uint32_t)(4 + (y & 3)) << 16);
      const dpk16 lo = __builtin_bit_cast(
          dpk16, __builtin_amdgcn_perm(LQ[y >> 2][2 * i + 1], LQ[y >> 2][2 * i], sel));
      sum[i] = __builtin_bit_cast(uint32_t, (dpk16)(T[y * 4 + i] + lo));
    }
    O[y * 2] = sat_pk_u8_hi(sat_pk_u8(sum[0]), sum[1]);
    O[y * 2 + 1] = sat_pk_u8_hi(sat_pk_u8(sum[2]), sum[3]);
  }
  (void)shift;
}

// One lane of a lane pair (lanes l and l + 32 of a wave: each half-wave then reads
// 32 adjacent symbol bytes of ONE plane per LDS instruction, free of bank
// conflicts): lane s transforms channels 2s and 2s+1 of tile u in block row v
// (packed int16 butterflies, see tile_plane), the pair swaps halves with one
// v_permlane32_swap per register pair, and lane s finishes pixel rows 4s..4s+3 --
// colour inverse on packed pairs (ycbcr.cpp:54-82), then two 16-byte stores per
// pixel row.  sym: the block row's symbols (channel c, scan index k, tile u at
// sym[(c*64 + k)*cols + u]), LDS or HBM; low: the frame's low-res planes.  Both
// lanes of a pair must be active (pair_tile / pair_half give the mapping).
// Item index -> (tile, half): 32 tiles per wave, lanes 0..31 half 0, lanes 32..63 half 1.
__device__ __forceinline__ int pair_tile(int it) { return (it >> 6) * 32 + (it & 31); }
__device__ __forceinline__ int pair_half(int it) { return (it >> 5) & 1; }

// FULL4: four channels, whole tiles only (W and H multiples of 8) -- the ragged-edge
// stores and the channel-count tests are compiled out.
template <int COLS, bool FULL4 = false>
__device__ __forceinline__ void transform_store_pair(const Geom &g, int cols_rt, const uint8_t *sym,
                                                     const uint8_t *low, const int16_t *s_unmap,
                                                     const uint8_t *s_shift, const uint32_t *s_shiftp,
                                                     int ycbcr, int u, int s, int v, uint8_t *img,
                                                     const uint32_t *pre_lr = nullptr, bool store_ok = true,
                                                     bool touched_on = false, uint32_t touched = 0) {
  const int cols = COLS > 0 ? COLS : cols_rt;
  const int C = FULL4 ? 4 : g.C;
  const int v2 = min(v + 1, g.rows - 1);
    const int u2 = min(u + 1, cols - 1);
    uint32_t QA[16], QB[16];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = 2 * s + cc;
      uint32_t O[16];
      if (FULL4 || c < C) {
        const uint8_t *m = low + (size_t)c * g.rows * cols;
        const int chroma = (ycbcr && (c == 1 || c == 2)) ? 1 : 0;  // decoder.cpp:376
        uint32_t lr0, lr8;
        if (pre_lr) {   // (inlined: a compile-time choice)
          lr0 = pre_lr[2 * cc]; lr8 = pre_lr[2 * cc + 1];
        } else {
          lr0 = (uint32_t)m[(size_t)v * cols + u] | ((uint32_t)m[(size_t)v * cols + u2] << 8);
          lr8 = (uint32_t)m[(size_t)v2 * cols + u] | ((uint32_t)m[(size_t)v2 * cols + u2] << 8);
        }
#if defined(HIMG_DEC_KO) && (HIMG_DEC_KO & 4)
        const int c_addr = cc;   // (both lanes of a pair read planes 0, 1: no bank shared between the half-waves)
#else
        const int c_addr = c;
#endif
        tile_plane<COLS>(sym + (size_t)c_addr * 64 * cols + u, cols, s_unmap, s_shift + chroma * 64,
                         s_shiftp + chroma * 32, lr0, lr8, O, COLS != 0 ? s_shiftp + 64 + 2 * chroma : nullptr);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) O[i] = 0;
      }
      if (cc == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) QA[i] = O[i];
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) QB[i] = O[i];
      }
    }
    // Swap halves between lane l and lane l + 32: one v_permlane32_swap per register
    // pair hands the upper lanes the partner's rows 4..7 of channels 0/1 and the
    // lower lanes the partner's rows 0..3 of channels 2/3 -- afterwards either half
    // holds its four pixel rows of all four channels, no selects.
    uint32_t ch0[8], ch1[8], ch2[8], ch3[8];   // [r*2+h]: pixel row 4s+r, x = 4h..4h+3
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const auto ra = __builtin_amdgcn_permlane32_swap(QA[i], QA[8 + i], false, false);
      const auto rb = __builtin_amdgcn_permlane32_swap(QB[i], QB[8 + i], false, false);
      ch0[i] = ra[0]; ch2[i] = ra[1];
      ch1[i] = rb[0]; ch3[i] = rb[1];
    }
    const int bw = FULL4 ? 8 : min(8, g.W - 8 * u);
    const int bh = FULL4 ? 8 : min(8, g.H - 8 * v);
    if (touched_on) {
      // (k_dec_row_fused's touch loads for the next row: issued before the gather, long landed --
      // their registers are given back here, in front of the pixel stores, so that nothing waits
      // for a store.)
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
      asm volatile("" :: "v"(touched));
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int y = 4 * s + rr;
      uint32_t px[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t q0 = ch0[rr * 2 + h], q1 = ch1[rr * 2 + h], q2 = ch2[rr * 2 + h], q3 = ch3[rr * 2 + h];
        if (ycbcr) {  // ycbcr.cpp:54-82, two pixels per packed op
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const uint32_t sel = k ? 0x0c030c02u : 0x0c010c00u;
            const dpk16 yy = __builtin_bit_cast(dpk16, __builtin_amdgcn_perm(0u, q0, sel));
            const dpk16 cbq = __builtin_bit_cast(dpk16, __builtin_amdgcn_perm(0u, q1, sel));
            const dpk16 crq = __builtin_bit_cast(dpk16, __builtin_amdgcn_perm(0u, q2, sel));
            const dpk16 c255 = {255, 255}, c254 = {254, 254}, one = {1, 1};
            // cbv = 2 cb - 255, crv = 2 cr - 255;  (cbv + crv + 2) >> 2 == (cb + cr - 254) >> 1 exactly.
            const dpk16 gg = yy - ((cbq + crq - c254) >> one);
            // r = g + crv, b = g + cbv as ONE multiply-add each on top of g - 255 (v_pk_mad_i16:
            // three packed operations for the pair instead of six; everything stays far inside int16).
            const dpk16 g255 = gg - c255;
            const uint32_t rs = sat_pk_u8(pk_mad2(crq, g255));
            const uint32_t gs = sat_pk_u8(__builtin_bit_cast(uint32_t, gg));
            const uint32_t bs = sat_pk_u8(pk_mad2(cbq, g255));
            const uint32_t rg = __builtin_amdgcn_perm(gs, rs, 0x05010400u);            // r0 g0 r1 g1
            const uint32_t ba = __builtin_amdgcn_perm(q3, bs, k ? 0x07010600u : 0x05010400u);  // b0 a0 b1 a1
            px[4 * h + 2 * k] = __builtin_amdgcn_perm(ba, rg, 0x05040100u);
            px[4 * h + 2 * k + 1] = __builtin_amdgcn_perm(ba, rg, 0x07060302u);
          }
        } else {
          const uint32_t t0 = __builtin_amdgcn_perm(q1, q0, 0x05010400u), t1 = __builtin_amdgcn_perm(q1, q0, 0x07030602u);
          const uint32_t w0 = __builtin_amdgcn_perm(q3, q2, 0x05010400u), w1 = __builtin_amdgcn_perm(q3, q2, 0x07030602u);
          px[4 * h + 0] = __builtin_amdgcn_perm(w0, t0, 0x05040100u);
          px[4 * h + 1] = __builtin_amdgcn_perm(w0, t0, 0x07060302u);
          px[4 * h + 2] = __builtin_amdgcn_perm(w1, t1, 0x05040100u);
          px[4 * h + 3] = __builtin_amdgcn_perm(w1, t1, 0x07060302u);
        }
      }
#if defined(HIMG_DEC_KO) && (HIMG_DEC_KO & 16)
      asm volatile("" :: "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(px[4]), "v"(px[5]), "v"(px[6]), "v"(px[7]));   // (computed, not stored)
      if (false) {
#else
      if (store_ok && (FULL4 || y < bh)) {
#endif
        uint8_t *dst = img + ((size_t)(8 * v + y) * g.W + 8 * u) * C;
        if (FULL4 || (C == 4 && bw == 8)) {
          uint4 o0, o1;
          o0.x = px[0]; o0.y = px[1]; o0.z = px[2]; o0.w = px[3];
          o1.x = px[4]; o1.y = px[5]; o1.z = px[6]; o1.w = px[7];
          reinterpret_cast<uint4 *>(dst)[0] = o0;
          reinterpret_cast<uint4 *>(dst)[1] = o1;
        } else {
#pragma unroll
          for (int x = 0; x < 8; ++x)
            if (x < bw)
              for (int c = 0; c < C; ++c) dst[x * C + c] = (uint8_t)(px[x] >> (8 * c));
        }
      }
    }
}

// tile_plane's identity test, by one wavefront (l = 0..63).  n = the largest code with
// fmap[i] == i for all i <= n (mapper.h:33-35: a code unmaps to itself there);
// B = the largest power of two <= n, lowered until B << (largest shift of group H:
// rows 2..7 of columns 1..7) <= 2048 (inside both range conditions of the packed
// transform, packed_wht_exact).  dst[2 * chroma + {0, 1}]: (B, B), then the mask of the
// bits that a sum code + B outside [0, 2B) sets.  No usable range: the test always fails.
__device__ __forceinline__ void identity_test_words(const DecFrame *df, int l, uint32_t *dst) {
  const unsigned long long ne = __ballot(df->fmap[l] != l), ne2 = __ballot(df->fmap[64 + l] != 64 + l);
  const int n = ne ? __ffsll((long long)ne) - 2 : (ne2 ? 62 + __ffsll((long long)ne2) : 127);
  if (l < 2) {
    int smax = 0;
    for (int y = 2; y < 8; ++y)
      for (int x = 1; x < 8; ++x) smax = max(smax, (int)df->shift[l][y * 8 + x]);
    int B = 0;
    if (n >= 1) {
      B = 1 << (31 - __clz(n));
      while (B && ((long long)B << smax) > 2048) B >>= 1;
    }
    const uint32_t b16 = B ? (uint32_t)B : 0x4000u, m16 = B ? (uint32_t)(0xffffu & ~(2u * B - 1u)) : 0xffffu;
    dst[2 * l] = b16 | (b16 << 16);
    dst[2 * l + 1] = m16 | (m16 << 16);
  }
}

// k_tile_inv: the transform of the unfused path (rows too wide for LDS): the
// symbols come from HBM (ws.fres_sym, written by k_dec_huff); two lanes per tile,
// 128 tiles per workgroup.
// FAST: four channels, whole tiles (tile_plane's identity-range gather, no ragged edges).
template <bool FAST>
__global__ __launch_bounds__(256) void k_tile_inv(Geom g, DecWs ws, uint8_t *out_frames, int v0) {
  __shared__ int16_t s_unmap[256];   // indexed by the code byte
  __shared__ uint8_t s_shift[2][64];
  __shared__ uint32_t s_shiftp[2 * 32 + 4];   // [chroma][register pair], then the identity-test words
  const int v = blockIdx.y + v0, f = blockIdx.z;
  const DecFrame *df = ws.frames + f;
  __shared__ int s_status;
  if (threadIdx.x == 0) s_status = df->status;
  __syncthreads();
  if (s_status) return;
  {
    const int k = threadIdx.x;
    const int sc = (int8_t)k;
    s_unmap[k] = (int16_t)(sc >= 0 ? df->fmap[sc] : (sc == -128 ? -df->fmap[127] : -df->fmap[-sc]));
    if (k < 128) s_shift[k >> 6][k & 63] = df->shift[k >> 6][k & 63];
    if (k < 64) {
      const int ch = k >> 5, e = k & 31, x = e >> 2, j = e & 3;
      s_shiftp[ch * 32 + e] = (uint32_t)df->shift[ch][(2 * j) * 8 + x] | ((uint32_t)df->shift[ch][(2 * j + 1) * 8 + x] << 16);
    }
    if (FAST && k >= 192) identity_test_words(df, k - 192, s_shiftp + 64);
  }
  __syncthreads();
  const int it = blockIdx.x * 256 + threadIdx.x;   // (tile, half): both lanes of a pair are in or out
  if (pair_tile(it) >= g.cols) return;
  transform_store_pair<(FAST ? -1 : 0), FAST>(g, g.cols, ws.fres_sym + (size_t)f * ws.fres_stride + (size_t)v * g.row_block,
                          ws.low + (size_t)f * ws.plane_stride, s_unmap, &s_shift[0][0], s_shiftp,
                          df->ycbcr, pair_tile(it), pair_half(it), v,
                          out_frames + (size_t)f * ((size_t)g.W * g.H * g.C));
}

// COLS > 0 fixes the tile count per block row at compile time (512 = 4096-pixel
// rows; COLS < 0: any width of four-channel whole tiles, run-time count, the same transform paths): the 64 symbol slots of a tile are then at immediate LDS offsets instead
// of 64 live address registers, which is what keeps the transform phase from
// spilling.
// A workgroup takes `rpw` consecutive block rows: narrow rows leave most of the 1024
// lanes without a tile in the transform (two lanes per tile: 480 lanes at 1920
// pixels), so as many rows as fit the LDS -- and the lanes -- are entropy-decoded one
// after the other (each by all 1024 lanes) and then transformed together.
template <int COLS>
__device__ __forceinline__ void dec_row_fused_body(const Geom &g, const DecWs &ws, const uint8_t *packed,
                                                   size_t in_stride, const uint32_t *sizes,
                                                   uint8_t *out_frames, int r0, int r1, int rpw,
                                                   const int bx, const int f, const int gx, const int gy,
                                                   const uint32_t next_lin, const bool again, const bool same_tables) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (COLS == 512) rpw = 1;   // 4096-pixel rows: one row fills the lanes and the LDS (known at compile time)
  const FusedLayout L = fused_layout(g.row_block, rpw);
  const uint32_t rb16 = ((uint32_t)g.row_block + 15u) & ~15u;   // one row's symbols
  uint8_t *sym0 = smem + L.sym;
  LdsTables &T = *reinterpret_cast<LdsTables *>(smem + L.tab);
  StreamShared *sh = reinterpret_cast<StreamShared *>(smem + L.sh);
  int16_t *s_unmap = reinterpret_cast<int16_t *>(smem + L.unmap);
  uint8_t *s_shift = smem + L.shift;
  uint32_t *s_shiftp = reinterpret_cast<uint32_t *>(smem + L.shiftp);

  // (opaque per row: what depends on the lane alone would otherwise be computed once in front of
  // the persistent loop and stay in registers across the whole body)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  __builtin_assume(tid_ >= 0 && tid_ < kDecThreads);   // (the range the compiler knew of threadIdx.x)
  const int tid = tid_;
  const long long c_in = clock64();
  DecFrame *df = ws.frames + f;
  const uint8_t *low = ws.low + (size_t)f * ws.plane_stride;
  // 4096-pixel rows: one row per workgroup, every lane knows its tile from the start.
  // Everything the row needs from global memory is requested HERE, in front of the
  // first barrier -- the frame's verdict, k_row_count's lane record, the row index,
  // the low-res corners of the lane's two planes, the tables -- so that the round
  // trips overlap instead of following each other (each is 1-2 us of a 43 us row).
  PreLane pl = {};
  uint32_t pre_lr[4] = {0, 0, 0, 0}, pre_off0 = 0, pre_len0 = 0;
  if constexpr (COLS == 512) {
    const size_t ri = (size_t)f * g.rows + (size_t)(r0 + bx);
    const uint32_t *ps = ws.lane_start + ri * kDecThreads, *po = ws.lane_off + ri * (kDecThreads + kRecHdr);
    pl.start = ps[tid]; pl.off = po[tid]; pl.nxt = po[tid + 1];
    pl.nstart = tid + 1 < kDecThreads ? ps[tid + 1] : ~0u;
    pl.tot = po[kDecThreads]; pl.endrel = po[kDecThreads + 1]; pl.valid = po[kDecThreads + 2]; pl.rounds = po[kDecThreads + 3];
    pre_off0 = ws.row_off[ri]; pre_len0 = ws.row_len[ri];
    const int u = pair_tile(tid), hs = pair_half(tid), v = r0 + bx;
    const int u2 = min(u + 1, COLS - 1), v2 = min(v + 1, g.rows - 1);
    // (32-bit offsets from the frame's plane: a low-res plane set is C * rows * cols bytes)
    const uint32_t pstride = (uint32_t)g.rows * COLS, o1 = (uint32_t)v * COLS, o2 = (uint32_t)v2 * COLS;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const uint32_t m = (uint32_t)(2 * hs + cc) * pstride;
      pre_lr[2 * cc] = (uint32_t)low[m + o1 + u] | ((uint32_t)low[m + o1 + u2] << 8);
      pre_lr[2 * cc + 1] = (uint32_t)low[m + o2 + u] | ((uint32_t)low[m + o2 + u2] << 8);
    }
  }
  // The frame's verdict: ONE read for the workgroup (other workgroups flag the frame
  // concurrently).  4096-pixel rows: requested here, stored with the tables below -- a
  // store right here would make wave 0 wait for every load above before it requests
  // its share of the tables, two round trips in front of the first barrier.
  uint32_t st0 = 0;
  if (tid == 0) st0 = df->status;
  if constexpr (COLS != 512) {
    if (tid == 0) sh->flag = st0;
    __syncthreads();
    if (sh->flag) return;
  }
  const uint8_t *p = packed + (size_t)f * in_stride;
  // The decode tables: the loads first (one group-table and one sub-table quad-word
  // per lane, a tree node for the first 522 -- every slot of the workspace, so that the
  // node count need not be known), then the clear of the symbol area below while they
  // are in flight, then the stores.
  static_assert((1 << kLutBits) / 2 == kDecThreads && kSubEntries / 2 <= kDecThreads && kMaxNodes + 1 <= kDecThreads,
                "one table element per lane");
  // (A persistent workgroup whose last row was of the same frame still holds them.)
  const bool ld_tab = !(COLS == 512 && same_tables);
  uint4 t_grp = make_uint4(0, 0, 0, 0), t_sub = t_grp, t_row = t_grp;
  uint32_t t_nd = 0;
  static_assert(kRowTabWords / 4 <= 64, "one wavefront copies them");
  if (ld_tab) {
    t_grp = reinterpret_cast<const uint4 *>(ws.grp + ((size_t)f * 2 + 1) * (1u << kLutBits))[tid];
    if (tid < kSubEntries / 2) t_sub = reinterpret_cast<const uint4 *>(ws.sub + ((size_t)f * 2 + 1) * kSubEntries)[tid];
    if (tid < kMaxNodes + 1) t_nd = (ws.nodes + ((size_t)f * 2 + 1) * (kMaxNodes + 1))[tid];
    // The transform's tables: one 16-byte word per lane of wave 0 from the frame's copy (k_dec_parse).
    if (tid < kRowTabWords / 4) t_row = reinterpret_cast<const uint4 *>(df->row_tabs)[tid];
  }
  const GrpTables tb = tables_of(&T);
  const int rb = r0 + bx * rpw;
  const int nr = COLS == 512 ? 1 : min(rpw, r1 - rb);
  // A persistent workgroup's next row: the loads above are in flight while the slowest wavefront
  // still transforms the row before; nothing of the LDS is written before it is done.
  if (again) __syncthreads();
  {
    uint4 z;
    z.x = z.y = z.z = z.w = 0;
    const int n16 = (int)(rb16 >> 4) * nr;
    for (int k = tid; k < n16; k += kDecThreads) reinterpret_cast<uint4 *>(sym0)[k] = z;
  }
  if (ld_tab) {
    reinterpret_cast<uint4 *>(T.grp)[tid] = t_grp;
    if (tid < kRowTabWords / 4) reinterpret_cast<uint4 *>(smem + L.unmap)[tid] = t_row;   // unmap, shift, shiftp: contiguous
    if (tid < kSubEntries / 2) reinterpret_cast<uint4 *>(T.grp + (1 << kLutBits))[tid] = t_sub;
    if (tid < kMaxNodes + 1) T.nd[tid] = t_nd;
  }
  if constexpr (COLS == 512) {
    if (tid == 0) { sh->flag = st0; sh->err = 0; sh->endbit = ~0ull; }   // (decode_stream<.., PRIMED>)
  }
  __syncthreads();
  if constexpr (COLS == 512) {
    if (sh->flag) return;   // (uniform: one read, in front of the barrier above)
  }

  auto decode_row = [&](int i) {
    const int r = rb + i;
    if (tid == 0) {
      uint32_t *st = ws.stats + ((size_t)f * (g.rows + 1) + r + 1) * 8;
      st[2] = 0; st[3] = 0;   // atomicMax targets, see the end of the kernel
    }
    if constexpr (COLS == 512) {
      uint32_t *st = ws.stats + ((size_t)f * (g.rows + 1) + r + 1) * 8;
      const int rc = decode_row_recorded(p, sizes[f], pre_off0, pre_len0, (uint32_t)g.row_block, tb, sh, sym0, st, pl);
      if (rc >= 0) return rc;
      return decode_stream_fused_cold(p, sizes[f], pre_off0, pre_len0, (uint32_t)g.row_block,
                                      reinterpret_cast<const uint32_t *>(tb.grp), tb.nd, sh, sym0, st,
                                      (uint32_t)g.max_sub, (uint32_t)g.lead_bits);
    }
    // Other widths: the same two paths, the lane's record read here (rows of a workgroup are
    // decoded one after the other: the loads of row i + 1 cannot be in flight under the prologue).
    const size_t ri = (size_t)f * g.rows + (size_t)r;
    const uint32_t *ps = ws.lane_start + ri * kDecThreads, *po = ws.lane_off + ri * (kDecThreads + kRecHdr);
    PreLane q;
    q.start = ps[tid]; q.off = po[tid]; q.nxt = po[tid + 1];
    q.nstart = tid + 1 < kDecThreads ? ps[tid + 1] : ~0u;
    q.tot = po[kDecThreads]; q.endrel = po[kDecThreads + 1]; q.valid = po[kDecThreads + 2]; q.rounds = po[kDecThreads + 3];
    const uint32_t off_r = ws.row_off[ri], len_r = ws.row_len[ri];
    // The row before ends by reading sh->err / sh->endbit with no barrier behind the reads: a slow
    // wavefront must not meet this row's reset while it still decides that one.
    if (i > 0) __syncthreads();
    if (tid == 0) { sh->err = 0; sh->endbit = ~0ull; }
    __syncthreads();
    uint32_t *st = ws.stats + ((size_t)f * (g.rows + 1) + r + 1) * 8;
    const int rc = decode_row_recorded(p, sizes[f], off_r, len_r, (uint32_t)g.row_block, tb, sh, sym0 + (size_t)i * rb16, st, q);
    if (rc >= 0) return rc;
    return decode_stream_fused_cold(p, sizes[f], off_r, len_r, (uint32_t)g.row_block,
                                    reinterpret_cast<const uint32_t *>(tb.grp), tb.nd, sh, sym0 + (size_t)i * rb16, st,
                                    (uint32_t)g.max_sub, (uint32_t)g.lead_bits);
  };
  int bad = 0;
  if constexpr (COLS == 512) {
    bad = decode_row(0);
  } else {
#pragma unroll 1
    for (int i = 0; i < nr && !bad; ++i) bad = decode_row(i);
  }
  if (bad) {   // uniform: every lane gets the same verdict
    if (tid == 0) atomicMax(&df->status, fmt_err(7, 1));
    return;
  }

  const long long c_p2 = clock64();
  // What the workgroup that takes this CU next will ask for first -- the packed bytes of the row 256
  // workgroups on (workgroups go to the XCDs round robin, so that one shares this one's L2) -- is
  // touched now, a line per lane of wavefronts 1..5: its first steps then wait for the L2 instead of
  // the HBM (decode of 128 frames 10.13 -> 9.95 ms; touching that row's lane record as well gains
  // nothing).  Loads nobody reads: the register is kept out of the compiler's hands until the
  // explicit wait in front of the pixel stores (transform_store_pair).
  uint32_t pf_a = 0;
  bool pf_on = false;
  {
    // (next_lin: the grid element this workgroup -- or its successor on this CU -- takes next)
    const int fn = (int)(next_lin / (uint32_t)gx), rn = r0 + (int)(next_lin % (uint32_t)gx) * rpw;
    pf_on = g.prefetch_rows != 0 && fn < gy && rn < r1;
    if (pf_on) {
      const int wv = tid >> 6, ln = tid & 63;
      // (how many bytes: this workgroup's own rows are the best guess -- 4096 pixels: the row's length
      // is at hand; other widths: the frame's mean)
      const uint32_t guess = COLS == 512 ? pre_len0 : (uint32_t)((unsigned long long)sizes[f] * (uint32_t)nr / (uint32_t)g.rows);
      const uint32_t o = 128u * (uint32_t)((wv - 1) * 64 + ln);
      if (wv >= 1 && wv <= 5 && o < guess + 256u) {
        // Never beyond that frame's stream: its rows may be shorter than this one's, and a frame
        // that failed to parse has no index at all (whatever the slot holds is compared, not trusted).
        const unsigned long long at = (unsigned long long)ws.row_off[(size_t)fn * g.rows + (size_t)rn] + o;
        if (at + 4ull <= (unsigned long long)sizes[fn]) {
          const uint8_t *a = packed + (size_t)fn * in_stride + at;
          const uint8_t *al = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(a) & ~(uintptr_t)3);
          asm volatile("global_load_dword %0, %1, off" : "=v"(pf_a) : "v"(al) : "memory");
        }
      }
    }
  }
  const int ycbcr = df->ycbcr;
  const int cols = COLS > 0 ? COLS : g.cols;
  // ---- phase 2: inverse transform, colour inverse and stores ----
  // Two adjacent lanes share a tile (transform_store_pair).  The decoded symbols
  // stay read-only in LDS: no barrier, no second pass over them.
  uint8_t *img = out_frames + (size_t)f * ((size_t)g.W * g.H * g.C);
  const int per_row = ((cols + 31) >> 5) * 64;   // whole wavefronts: a lane pair never straddles rows
  HIMG_SPAN_BEGIN("dec.transform");
#pragma unroll 1
  for (int it = tid; it < per_row * nr; it += kDecThreads) {
    const int i = COLS == 512 ? 0 : it / per_row, il = it - i * per_row;
    // A lane pair beyond the row's last tile (widths that are not a multiple of 256 pixels)
    // transforms the last tile once more and stores nothing: no exec-masked region around
    // the transform (with compile-time strides it cost 50 VGPR spills at 1920 pixels).
    const bool in_row = pair_tile(il) < cols;
    transform_store_pair<COLS, COLS != 0>(g, cols, sym0 + (size_t)i * rb16, low, s_unmap, s_shift, s_shiftp, ycbcr,
                                            in_row ? pair_tile(il) : cols - 1, pair_half(il), rb + i, img,
                                            COLS == 512 ? pre_lr : nullptr, in_row, pf_on && it == tid, pf_a);
  }
  HIMG_SPAN_END("dec.transform");
  // Cycle stamps: the slowest wave counts (the SIMDs issue oldest-first, so the
  // first wave finishes long before the last one).
  if ((tid & 63) == 0) {
    const long long c_out = clock64();
    for (int i = 0; i < nr; ++i) {
      uint32_t *st = ws.stats + ((size_t)f * (g.rows + 1) + rb + i + 1) * 8;
      atomicMax(&st[2], (uint32_t)((c_out - c_p2) >> 4));   // transform + stores
      atomicMax(&st[3], (uint32_t)((c_out - c_in) >> 4));   // the whole workgroup
    }
  }
}


// The kernel: a workgroup takes the rows bx = blockIdx.x, blockIdx.x + gridDim.x, ... of the
// launch's (rows / rpw) x batch grid, frame after frame.  With as many workgroups as the GPU
// holds at once (launch_decode: one per CU) a CU does not wait for a new workgroup -- its LDS, its
// wavefronts, their arguments -- between two rows (~3.6 k of a row's 61 k cycles), and the next
// row's records are requested while the last wavefronts of this row still transform.
// A kernel-argument struct into registers, word by word (a struct in the constant address space has
// no copy constructor the host pass accepts).
template <class W, class T>
__device__ __forceinline__ void karg_copy(T *dst, const __attribute__((address_space(4))) T *src) {
  static_assert(sizeof(T) % sizeof(W) == 0 && alignof(T) >= alignof(W), "whole words");
  const __attribute__((address_space(4))) W *sw = (const __attribute__((address_space(4))) W *)src;
  W *dw = reinterpret_cast<W *>(dst);
#pragma unroll
  for (size_t i = 0; i < sizeof(T) / sizeof(W); ++i) dw[i] = sw[i];
}
struct RowArgs {
  Geom g;
  DecWs ws;
  const uint8_t *packed;
  size_t in_stride;
  const uint32_t *sizes;
  uint8_t *out_frames;
  int r0, r1, rpw, gx, gy, next_dist;
};
// (The arguments are read from the kernel-argument segment anew for every row, through a pointer
// the compiler cannot see through: read once in front of the loop they all stay live across the
// body -- 104 scalar registers spilled instead of 36, the row kernel 9 % slower.)
template <int COLS>
__global__ __launch_bounds__(kDecThreads) void k_dec_row_fused(RowArgs) {
  typedef const __attribute__((address_space(4))) RowArgs *KArgs;
  KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t total = (uint32_t)ka->gx * (uint32_t)ka->gy, S = gridDim.x;
  // Round i: the grid elements [i S, (i + 1) S), this workgroup the one at (blockIdx.x + 37 i) mod S
  // -- rotated, or a workgroup would meet the same row positions of every frame (S = 256, 512 rows
  // per frame: two of them) and the slow rows of similar frames would all be one workgroup's:
  // 128 identical frames 10.28 ms against 10.01 with a workgroup per row.
  auto element = [&](uint32_t i) { return i * S + (blockIdx.x + 37u * i) % S; };
  bool again = false;
  int f_prev = -1;
#pragma unroll 1
  for (uint32_t i = 0; i * S < total; ++i) {
    const uint32_t lin = element(i);
    if (lin >= total) break;   // (the last round is a partial one)
    asm volatile("" : "+s"(ka));
    Geom g;
    DecWs ws;
    karg_copy<uint32_t>(&g, &ka->g);
    karg_copy<unsigned long long>(&ws, &ka->ws);
    const int gx = ka->gx, f = (int)(lin / (uint32_t)gx);
    const uint32_t nd = (uint32_t)ka->next_dist;
    dec_row_fused_body<COLS>(g, ws, ka->packed, ka->in_stride, ka->sizes, ka->out_frames, ka->r0, ka->r1, ka->rpw,
                             (int)(lin % (uint32_t)gx), f, gx, ka->gy, nd ? lin + nd : element(i + 1), again, f == f_prev);
    again = true;
    f_prev = f;
  }
}

// ---------------------------------------------------------------------------
// k_row_count: the fixpoint rounds of every FRES block row, on their own.  They
// need no symbol storage (tables + the row's payload instead of the row's 128 KiB
// of symbols), so two workgroups share a CU at 8 waves per SIMD.  Output per lane:
// its first owned token and the exclusive prefix of the symbol counts; the fused
// kernel then goes straight to its write pass.
//   * a workgroup walks kRowsPerCount consecutive rows of one frame: the 28 KiB of
//     decode tables are loaded once for all of them (they were 17 % of a
//     one-row workgroup's cycles);
//   * the row's payload is staged in LDS (coalesced 4-byte loads) and the lanes'
//     bit windows refill from there;
//   * nothing is fenced at the end: the consumer is a later kernel.  (The
//     release fence + barrier that used to close the kernel wrote back L2 and cost
//     20 % of its duration.)
// ---------------------------------------------------------------------------
constexpr int kRowsPerCount = 4;

// Exclusive scan of a 32-bit value over the 1024-thread workgroup (DPP-free:
// 6 shuffles + one LDS exchange).
__device__ __forceinline__ uint32_t block_scan_u32d(uint32_t v, uint32_t *sm, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) sm[wave] = incl;
  __syncthreads();
  uint32_t pre = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kDecThreads / 64; ++w) {
    const uint32_t x = sm[w];
    if (w < wave) pre += x;
    tot += x;
  }
  *total = tot;
  return pre + incl - v;
}

// QTR: three more boundaries inside every lane's range (lean_fixpoint's marks at 1/4, 1/2,
// 3/4) go to l_q ([6][kDecThreads]: positions, then output offsets), and the quarter
// records that hold the first symbol of every 128 KiB window of the row's symbols to the
// header (k_row_window then has 1024 sub-sequences per window to walk instead of 256).
template <class RD, bool QTR = false>
__device__ __forceinline__ void row_count_one(RD &rd, const GrpTables &tb, StreamShared *sh, uint32_t *sm32,
                                              uint32_t rel0, uint32_t rem, uint32_t sb, uint32_t lead_bits,
                                              uint32_t *l_start, uint32_t *l_off, uint32_t *rc, long long c_in,
                                              uint32_t *l_q = nullptr, uint32_t out_size = 0, uint32_t window = 0) {
  const int tid = threadIdx.x;
  const uint32_t rel_end = rel0 + rem;
  const SubGrid sg = sub_grid(rel0, rem, sb, tid);
  const uint32_t my_b0 = sg.b0, lim = sg.lim;
  const bool active = sg.active;
  const int last_active = sg.last_active;
  uint32_t start = active ? my_b0 : rel_end, endpos = start, cnt = 0, rounds = 0;
  long long c_phase[2] = {0, 0};
  const long long c_fix0 = clock64();
  uint32_t mark[3] = {0, 0, 0}, mcnt[3] = {0, 0, 0};
  if (QTR) {
    const uint32_t q = active ? (lim - my_b0) >> 2 : 0u;
    mark[0] = start + q; mark[1] = start + 2u * q; mark[2] = start + 3u * q;
    lean_fixpoint<true, RD, 0, 3>(rd, tb, sh, rel0, active, lim, &start, &endpos, &cnt, &rounds, false, lead_bits, nullptr,
                                  c_phase, nullptr, mark, mcnt);
  } else {
    lean_fixpoint<true>(rd, tb, sh, rel0, active, lim, &start, &endpos, &cnt, &rounds, false, lead_bits, nullptr, c_phase);
  }
  const long long c_fix1 = clock64();
  // 32-bit scan: a lane's count is clamped to 2^22 - 1, more than a whole row holds
  // (the caller only comes here for rows below 2^22 symbols), so 1024 of them cannot
  // wrap, valid streams are exact, and a lane that claims more still overruns the
  // block in the write pass and is rejected there like before.
  uint32_t tot;
  const uint32_t cc = min(cnt, 0x3fffffu);
  const uint32_t off = block_scan_u32d(cc, sm32, &tot);
  l_start[tid] = start - rel0;
  l_off[tid] = off;
  if (QTR) {
    uint32_t o[5];
    o[0] = off; o[4] = off + cc;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      o[k + 1] = off + min(mcnt[k], cc);
      l_q[k * kDecThreads + tid] = mark[k] - rel0;
      l_q[(3 + k) * kDecThreads + tid] = o[k + 1];
    }
    // The quarter record that holds the first symbol of window w (w >= 1).
    if (window && cc) {
      for (uint32_t w = (o[0] + window - 1u) / window; w * window < o[4] && w * window < out_size && w < (uint32_t)(kRecHdr - kRecWin); ++w) {
        if (w == 0) continue;
        const uint32_t X = w * window;
        int k = 0;
#pragma unroll
        for (int j = 1; j < 4; ++j) if (X >= o[j]) k = j;
        l_off[kDecThreads + kRecWin + w] = 4u * (uint32_t)tid + (uint32_t)k;
      }
    }
  }
  if (tid == last_active) l_off[kDecThreads + 1] = endpos - rel0;
  if (tid == 0) {
    l_off[kDecThreads] = tot;
    l_off[kDecThreads + 3] = rounds | (min(sh->dbg[0], 4095u) << 8) | (min(sh->dbg[1], 4095u) << 20);
    l_off[kDecThreads + 4] = QTR ? 1u : 0u;
    l_off[kDecThreads + 2] = 1;   // (no fence: the consumer is a later kernel)
  }
  if ((tid & 63) == 0 && rc) {   // cycles / 16, slowest wave of the workgroup
    atomicMax(&rc[0], (uint32_t)((c_fix0 - c_in) >> 4));            // tables / payload staging
    atomicMax(&rc[1], (uint32_t)(c_phase[0] >> 4));                 // lead-in
    atomicMax(&rc[2], (uint32_t)(c_phase[1] >> 4));                 // ... to the end of round 1
    atomicMax(&rc[3], (uint32_t)((c_fix1 - c_fix0) >> 4));          // the whole fixpoint
    atomicMax(&rc[4], (uint32_t)((clock64() - c_in) >> 4));         // the row
  }
}

// LDSPAY: the row's payload is staged in LDS (rows of up to 36 KiB -- every shape
// the fused row kernel serves); otherwise (wide rows, generic path) the lanes read
// it in place.
template <bool LDSPAY>
__global__ __launch_bounds__(kDecThreads, 8) void k_row_count(Geom g, DecWs ws, const uint8_t *packed,
                                                           size_t in_stride, const uint32_t *sizes,
                                                           int r0, int r1, int rows_per_wg) {
  // Both tables as two arrays: the step words (gy), then their .x words (gx).
  __shared__ __attribute__((aligned(16))) uint32_t gyx[2 * kTabEntries];
  __shared__ __attribute__((aligned(16))) uint32_t s_pay[LDSPAY ? kPayWords : 4];   // the row's payload
  __shared__ uint32_t nd[kMaxNodes + 1];
  uint32_t *gy = gyx, *gx = gyx + kTabEntries;
  __shared__ uint32_t sm32[kDecThreads / 64];
  __shared__ StreamShared sh;
  const int f = blockIdx.y, tid = threadIdx.x;
  DecFrame *df = ws.frames + f;
  // One read for the whole workgroup: the LRES kernels run concurrently on the
  // other stream and may flag the frame while this kernel starts.
  if (tid == 0) {
    // The row walk ran beside k_dec_parse: its verdict counts if the parse passed.
    const int w = df->parse_status == 0 ? df->walk_status : 0;
    if (w && blockIdx.x == 0) atomicMax(&df->status, w);
    sh.flag = df->status | w;
  }
  __syncthreads();
  const int failed = sh.flag;
  if (!failed) {   // load_dec_tables with the count-only step words next to the long-code descriptors
    const uint32_t *nodes = ws.nodes + ((size_t)f * 2 + 1) * (kMaxNodes + 1);
    const int nn = df->s[1].num_nodes;
    for (int k = tid; k < nn; k += kDecThreads) nd[k] = nodes[k];
    const uint4 *gg = reinterpret_cast<const uint4 *>(ws.grp + ((size_t)f * 2 + 1) * (1u << kLutBits));
    const uint2 *gc = reinterpret_cast<const uint2 *>(ws.gyc + ((size_t)f * 2 + 1) * (1u << kLutBits));
    for (int k = tid; k < (1 << kLutBits) / 2; k += kDecThreads) {
      const uint4 q = gg[k];
      reinterpret_cast<uint2 *>(gx)[k] = make_uint2(q.x, q.z);   // long-code descriptors
      reinterpret_cast<uint2 *>(gy)[k] = gc[k];                  // count-only step words
    }
    const uint4 *gs = reinterpret_cast<const uint4 *>(ws.sub + ((size_t)f * 2 + 1) * kSubEntries);
    for (int k = tid; k < kSubEntries / 2; k += kDecThreads) {
      const uint4 q = gs[k];
      reinterpret_cast<uint2 *>(gx + (1 << kLutBits))[k] = make_uint2(q.x, q.z);
      reinterpret_cast<uint2 *>(gy + (1 << kLutBits))[k] = make_uint2(q.y, q.w);
    }
  }
  GrpTables tb;
  tb.grp = nullptr; tb.gx = gx; tb.gy = gy; tb.nd = nd;
  const uint8_t *p = packed + (size_t)f * in_stride;
  const int rb = r0 + (int)blockIdx.x * rows_per_wg;
  for (int r = rb; r < min(rb + rows_per_wg, r1); ++r) {
    const long long c_in = clock64();
    uint32_t *l_start = ws.lane_start + ((size_t)f * g.rows + r) * kDecThreads;
    uint32_t *l_off = ws.lane_off + ((size_t)f * g.rows + r) * (kDecThreads + kRecHdr);
    uint32_t *rc = ws.rc_stats ? ws.rc_stats + ((size_t)f * g.rows + r) * 8 : nullptr;
    if (tid == 0) { l_off[kDecThreads + 2] = 0; sh.dbg[0] = sh.dbg[1] = 0; }   // not usable until proven otherwise
    if (tid >= kRecWin && tid < kRecHdr) l_off[kDecThreads + tid] = ~0u;            // no window index yet (k_row_window checks what it finds)
    const uint32_t pay_off = ws.row_off[(size_t)f * g.rows + r], pay_len = ws.row_len[(size_t)f * g.rows + r];
    const unsigned long long rem = 8ull * pay_len;
    uint32_t sb = (uint32_t)((rem + kDecThreads - 1) / kDecThreads);
    sb = (sb + 31u) & ~31u;
    sb = sb < kMinSubBits ? kMinSubBits : sb;
    // More than one chunk: the fused kernel does it all.
    if (failed || sb > (uint32_t)g.max_sub || rem == 0 || g.row_block >= (1 << 22)) continue;
    GReader rd;
    const uint32_t rel0 = rd.attach(p, sizes[f], 8ull * pay_off);
    if (LDSPAY) {
      const uint32_t nd = (rel0 + (uint32_t)rem + 31u) / 32u;   // dwords that hold payload bits
      // A payload beyond the staging buffer is left to the row kernels, like a row of
      // several chunks.
      if (nd + kPayPad > (uint32_t)kPayWords) continue;
      __syncthreads();   // the previous row's readers are done with s_pay (and the tables are in)
      stage_payload(rd, s_pay, nd + kPayPad);
      __syncthreads();
      LReader lr;
      lr.w = (const __attribute__((address_space(3))) uint32_t *)s_pay;
      lr.jmax = nd + kPayPad - 1u;
      row_count_one(lr, tb, &sh, sm32, rel0, (uint32_t)rem, sb, (uint32_t)g.lead_bits, l_start, l_off, rc, c_in);
    } else {
      __syncthreads();   // the tables are in / the previous row is done with the exchange slots
      if (ws.lane_q)     // rows that go through windows: four records per lane
        row_count_one<GReader, true>(rd, tb, &sh, sm32, rel0, (uint32_t)rem, sb, (uint32_t)g.lead_bits, l_start, l_off, rc, c_in,
                                     ws.lane_q + ((size_t)f * g.rows + r) * (6 * kDecThreads), (uint32_t)g.row_block, kRowWindow);
      else
        row_count_one(rd, tb, &sh, sm32, rel0, (uint32_t)rem, sb, (uint32_t)g.lead_bits, l_start, l_off, rc, c_in);
    }
  }
}

// ---------------------------------------------------------------------------
// k_row_count_w: the same records (every one of the 1024 lanes of a row kernel: its first
// owned token and the exclusive prefix of the symbol counts), computed by ONE WAVEFRONT
// per row -- for batches, where there are rows enough to fill the GPU that way.  The
// wavefront takes the row kernel's lanes 64 at a time (phase j: lanes 64 j .. 64 j + 63,
// a contiguous 2 KiB of the payload, read in place): speculative lead-in, count, then
// the chain is made exact among the 64 (a lane starts where its left neighbour ended:
// one DPP move per round instead of an LDS exchange and two workgroup barriers; a lane
// whose guess was wrong walks to its re-join point again, 64 bits), and the
// prefix of the counts is a wave scan on top of the phases before.  Lane 0 of a phase
// starts where lane 63 of the phase before ended: exact, like the first lane of a chunk.
// No staging of the payload (staging a phase's 2.3 KiB in LDS per wavefront was measured:
// no faster -- the kernel is bound by the decode steps themselves), no 1024-lane scans,
// sixteen rows share one load of the decode tables.
// ---------------------------------------------------------------------------
constexpr int kCountRowsW = kDecThreads / 64;   // rows per workgroup (one per wavefront)

// Wavefront scan / shift on the DPP path (one VALU operation per step where __shfl_up is a
// ds_bpermute with its address arithmetic, a wait and a select).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_from_d(uint32_t identity, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_add_dpp(uint32_t v) {   // inclusive; lane 63: the total
  v += dpp_from_d<0x111, 0xf>(0, v);   // row_shr:1
  v += dpp_from_d<0x112, 0xf>(0, v);   // row_shr:2
  v += dpp_from_d<0x114, 0xf>(0, v);   // row_shr:4
  v += dpp_from_d<0x118, 0xf>(0, v);   // row_shr:8
  v += dpp_from_d<0x142, 0xa>(0, v);   // row_bcast:15
  v += dpp_from_d<0x143, 0xc>(0, v);   // row_bcast:31
  return v;
}
// Lane l gets lane l - 1's v, lane 0 gets `first` (wave_shr:1).
__device__ __forceinline__ uint32_t wave_shr1_dpp(uint32_t first, uint32_t v) {
  return dpp_from_d<0x138, 0xf>(first, v);
}

// A piece of the payload staged in LDS, read ON DEMAND: a step fetches the 32 stream bits
// at its position with one ds_read2_b32 and one v_alignbit instead of keeping a 64-bit
// register window alive (the refill of ReaderT -- ten instructions with a 64-bit shift
// and a global load -- is executed by the whole wavefront whenever one of its 64 lanes
// runs low, i.e. on almost every step).  A step consumes at most kLutBits + 14 = 25 bits.
// Layout: the lanes of a wavefront read at a stride of one sub-sequence (8 dwords for
// 256 bits) and advance at about the same pace, which in a plain copy is an 8-way bank
// conflict on every read.  The piece is therefore stored in blocks of 33 dwords -- 32 of
// the payload, then a COPY of the next block's first dword -- so that dword w sits at
// w + w / 32 (the lane stride becomes co-prime with the 32 banks) and the pair
// (w, w + 1) is adjacent for every w.
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));   // two dwords at a dword-aligned address: ds_read2_b32
struct LdsBits {
  uint32_t base;   // LDS byte address of the dword that holds bit 0
  __device__ __forceinline__ uint32_t window(uint32_t pos) const {
    const uint32_t w = pos >> 5;
    const uint32_t a = base + ((w + (w >> 5)) << 2);
    const u32x2 x = *(const __attribute__((address_space(3))) u32x2_a4 *)(uintptr_t)a;
    return __builtin_amdgcn_alignbit(x.y, x.x, pos);   // (the hardware takes pos[4:0])
  }
};
constexpr uint32_t kStageSubBits = 320;                                    // longest sub-sequence the staged form takes
constexpr uint32_t kStageWords = 64u * kStageSubBits / 32u + 32u;          // payload dwords of one phase: 64 sub-sequences + slack
constexpr uint32_t kStageAlloc = (kStageWords + kStageWords / 32u + 3u) & ~3u;  // the same in blocks of 33
static_assert(kStageWords % 32 == 0, "whole blocks");

// 64 stream bits in a register pair, for walk_token (no refill: one token at most).
struct BitWin64 {
  unsigned long long win;
  __device__ __forceinline__ void consume(int n) { win >>= n; }
  __device__ __forceinline__ void refill() {}
};

// The GROUPS that start in [pos, lim), from the staged payload (lean_count<.., GRP = true>
// over an LdsBits): where the last one ends, how many symbols they produce.
__device__ __forceinline__ void grp_count_lds(const LdsBits &bits, const GrpTables &t, uint32_t pos, uint32_t lim,
                                              uint32_t *endpos, uint32_t *count) {
  uint32_t c = 0;
  if (pos < lim) {
    bool bad = false;
    const uint32_t TM = (uint32_t)kLutBits;   // (index BITS: the address is a v_bfe and a v_lshl_add)
    const uint32_t TB = lds_addr(t.gy);
    const uint32_t XOFF = (uint32_t)kTabEntries * 4u;   // from a step word to its .x
    uint32_t tm = TM, tb = TB;   // which table the next step indexes (lean_count: long codes take two steps)
    auto step = [&]() {
      uint32_t win = bits.window(pos);
      uint32_t a;   // (index << 2) + tb in ONE instruction (the compiler splits it: shift, add)
      asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(__builtin_amdgcn_ubfe(win, 0u, tm)), "v"(tb));
      uint32_t y = lds_ld32(a);
      uint32_t ntm = TM, ntb = TB;
      if (__builtin_expect(y == 0, 0)) {
        const uint32_t x = lds_ld32(a + XOFF);
        if ((x >> 31) && tm == TM) {
          ntm = x & 255u;
          ntb = TB + (((1u << kLutBits) + ((x >> 8) & 0xffffu)) << 2);
          y = (uint32_t)kLutBits | ((uint32_t)kLutBits << 27);
        } else {
          // Neither table resolves it: the tree walk, over a register window at pos.
          BitWin64 rd;   // 64 bits: the deepest code the walk follows is 32
          rd.win = (unsigned long long)win | ((unsigned long long)bits.window(pos + 32u) << 32);
          const int base = tm == TM ? 0 : kLutBits;
          uint32_t len, by;
          y = walk_token(rd, t, x, base, &len, &by, &bad);
          pos += len - (uint32_t)base;
          win = bits.window(pos);
        }
      }
      tm = ntm; tb = ntb;
      const uint32_t extra = __builtin_amdgcn_ubfe(win, y, y >> 5);
      pos += y >> 27;
      c += ((y >> 10) & 511u) + extra;
    };
    LoopCount lc;
    while (pos < lim) { HIMG_REGION_BEGIN("dec.count"); step(); lc.step(); HIMG_REGION_END("dec.count"); }
    lc.done(1);
    if (tm != TM) step();   // the loop ended between the two steps of a long code
  }
  *endpos = pos;
  *count = c;
}

// Rows whose sub-sequences are at most kStageSubBits bits long (4096-pixel rows up to
// ~2.4 bits per symbol): every phase's piece of the payload goes to LDS first -- coalesced
// 16-byte loads -- and the walks read it on demand (grp_count_lds); longer rows are read in
// place through register windows (wave-uniform choice per row).
__global__ __launch_bounds__(kDecThreads, 8) void k_row_count_w(Geom g, DecWs ws, const uint8_t *packed,
                                                               size_t in_stride, const uint32_t *sizes,
                                                               int r0, int r1) {
  __shared__ __attribute__((aligned(16))) uint32_t gyx[2 * kTabEntries];
  __shared__ uint32_t nd[kMaxNodes + 1];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[kCountRowsW * kStageAlloc];
  __shared__ int s_flag;
  uint32_t *gy = gyx, *gx = gyx + kTabEntries;
  const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  DecFrame *df = ws.frames + f;
  if (tid == 0) {
    // The row walk ran beside k_dec_parse: its verdict counts if the parse passed.
    const int w = df->parse_status == 0 ? df->walk_status : 0;
    if (w && blockIdx.x == 0) atomicMax(&df->status, w);
    s_flag = df->status | w;
  }
  __syncthreads();
  const int failed = s_flag;
  if (!failed) {   // load_dec_tables with the count-only step words next to the long-code descriptors
    const uint32_t *nodes = ws.nodes + ((size_t)f * 2 + 1) * (kMaxNodes + 1);
    const int nn = df->s[1].num_nodes;
    for (int k = tid; k < nn; k += kDecThreads) nd[k] = nodes[k];
    const uint4 *gg = reinterpret_cast<const uint4 *>(ws.grp + ((size_t)f * 2 + 1) * (1u << kLutBits));
    for (int k = tid; k < (1 << kLutBits) / 2; k += kDecThreads) {
      const uint4 q = gg[k];
      reinterpret_cast<uint2 *>(gx)[k] = make_uint2(q.x, q.z);   // bytes / long-code descriptors
      // The step words of the WRITE pass's groups (at most four output bytes), not the
      // count-only ones: a row kernel that walks those groups from a lane's recorded start
      // follows exactly this kernel's chain and lands on the next lane's start -- no
      // token-by-token tail there (1.3 % more steps here than with the longer groups).
      reinterpret_cast<uint2 *>(gy)[k] = make_uint2(q.y, q.w);
    }
    const uint4 *gs = reinterpret_cast<const uint4 *>(ws.sub + ((size_t)f * 2 + 1) * kSubEntries);
    for (int k = tid; k < kSubEntries / 2; k += kDecThreads) {
      const uint4 q = gs[k];
      reinterpret_cast<uint2 *>(gx + (1 << kLutBits))[k] = make_uint2(q.x, q.z);
      reinterpret_cast<uint2 *>(gy + (1 << kLutBits))[k] = make_uint2(q.y, q.w);
    }
  }
  __syncthreads();
  GrpTables tb;
  tb.grp = nullptr; tb.gx = gx; tb.gy = gy; tb.nd = nd;
  const int r = r0 + (int)blockIdx.x * kCountRowsW + (tid >> 6);
  if (r >= r1) return;
  const uint8_t *p = packed + (size_t)f * in_stride;
  uint32_t *l_start = ws.lane_start + ((size_t)f * g.rows + r) * kDecThreads;
  uint32_t *l_off = ws.lane_off + ((size_t)f * g.rows + r) * (kDecThreads + kRecHdr);
  if (lane == 0) l_off[kDecThreads + 2] = 0;   // not usable until proven otherwise
  const uint32_t pay_off = ws.row_off[(size_t)f * g.rows + r], pay_len = ws.row_len[(size_t)f * g.rows + r];
  const unsigned long long rem64 = 8ull * pay_len;
  uint32_t sb = (uint32_t)((rem64 + kDecThreads - 1) / kDecThreads);
  sb = (sb + 31u) & ~31u;
  sb = sb < kMinSubBits ? kMinSubBits : sb;
  // More than one chunk, or nothing to do: the row kernels do it all (k_row_count's rule).
  if (failed || sb > (uint32_t)g.max_sub || rem64 == 0 || g.row_block >= (1 << 22)) return;
  const uint32_t rem = (uint32_t)rem64;
  GReader rd;
  const uint32_t rel0 = rd.attach(p, sizes[f], 8ull * pay_off);
  const uint32_t rel_end = rel0 + rem;
  const uint32_t lead = (uint32_t)g.lead_bits;
  uint32_t first = rel0;   // where the phase's first lane starts: exact
  uint32_t base = 0;       // symbols in front of the phase
  uint32_t rounds = 0;
  uint32_t *stage = s_stage + (tid >> 6) * kStageAlloc;
  LdsBits bits;
  bits.base = lds_addr(stage);
  auto phases = [&](auto staged_c) {
  constexpr bool STAGED = decltype(staged_c)::value;
#pragma unroll 1
  for (int j = 0; j < kDecThreads / 64; ++j) {
    const int v = 64 * j + lane;
    const SubGrid q = sub_grid(rel0, rem, sb, v);
    const bool active = q.active;
    const uint32_t pb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.b0);   // the phase's first lane: its nominal start
    if (pb0 >= rel_end) {   // a phase beyond the payload: its lanes own nothing
      l_start[v] = rem;
      l_off[v] = base;
      continue;
    }
    // STAGED: positions below are relative to the dword `w0` of the reader's window, the
    // first one staged: the one that holds the first bit of the phase's first lane.
    uint32_t shift = 0;
    if constexpr (STAGED) {
      const uint32_t w0 = pb0 >> 5;
      shift = 32u * w0;
      wave_lds_sync();   // the walks of the phase before are done with the buffer
      for (uint32_t k = (uint32_t)lane; 4u * k < kStageWords; k += 64u) {
        const uint32_t w = w0 + 4u * k;
        uint4 x;
        if (w + 3u <= rd.jmax) {
          const PackedU4 u = *reinterpret_cast<const PackedU4 *>(rd.w + w);
          x.x = u.x; x.y = u.y; x.z = u.z; x.w = u.w;
        } else {
          x.x = rd.ld(w); x.y = rd.ld(w + 1u); x.z = rd.ld(w + 2u); x.w = rd.ld(w + 3u);
        }
        // Blocks of 33 (see LdsBits): four dwords of one block, and the block's first dword
        // once more behind the block before it.
        uint32_t *d = stage + 4u * k + (k >> 3);
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
        if ((k & 7u) == 0u && k) d[-1] = x.x;
      }
      wave_lds_sync();
    }
    auto walk = [&](uint32_t from, uint32_t to, uint32_t *e, uint32_t *c, bool cont) {
      if constexpr (STAGED) { (void)cont; grp_count_lds(bits, tb, from, to, e, c); }
      else lean_count<true, GReader, true>(rd, tb, from, to, e, c, cont);
    };
    const uint32_t b0 = q.b0 - shift, lim = q.lim - shift, lo0 = rel0 - shift, fst = first - shift;
    uint32_t start = active ? b0 : rel_end - shift;
    if (lane == 0 && active) start = fst;
    // Lead-in (see lean_fixpoint): a boundary of the token chain at or past the nominal
    // start, found from lead_bits in front of it.
    bool at_start = false;   // the reader stands at `start`
    if (lane > 0 && active && lead) {
      uint32_t from = start - lo0 > lead ? start - lead : lo0;
      if (from < pb0 - shift) from = pb0 - shift;   // (a lead-in longer than a sub-sequence: not in front of the phase)
      uint32_t guess, none;
      walk(from, start, &guess, &none, false);
      start = guess;
      at_start = true;
    }
    uint32_t endpos = start, cnt = 0;
    bool dirty = active;
    // Re-join (see lean_fixpoint): the walk is cut at T = nominal start + kJoinBits; a lane
    // whose start moves in a later round walks up to T again, and if it arrives at the
    // same boundary everything behind is what it already has.  A round after the first
    // then costs the wavefront kJoinBits instead of a whole sub-sequence.
    uint32_t T = (active ? b0 : rel_end - shift) + kJoinBits;
    if (T > lim || T < b0) T = lim;
    uint32_t posT = ~0u, cT = 0;
    for (;;) {
      if (dirty) {
        uint32_t p1, c1;
        walk(start, T, &p1, &c1, at_start);
        if (p1 == posT) {
          cnt = c1 + (cnt - cT);
        } else {
          uint32_t c2;
          walk(p1, lim, &endpos, &c2, start < T || at_start);
          cnt = c1 + c2;
        }
        posT = p1;
        cT = c1;
        at_start = false;
      }
      // The chain: a lane starts where its left neighbour ended (one DPP move).
      const uint32_t ns = wave_shr1_dpp(fst, endpos);
      dirty = active && ns != start;
      if (active) start = ns;
      ++rounds;
      if (!__any(dirty ? 1 : 0)) break;
    }
    // Exclusive prefix of the counts (k_row_count's clamp: see row_count_one).
    const uint32_t c = min(cnt, 0x3fffffu);
    const uint32_t incl = wave_scan_add_dpp(c);
    l_start[v] = start + shift - rel0;
    l_off[v] = base + incl - c;
    if (v == q.last_active) l_off[kDecThreads + 1] = endpos + shift - rel0;
    base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    // The next phase starts where this one's last lane ended (a phase of inactive lanes: nowhere).
    first = (uint32_t)__builtin_amdgcn_readlane((int)(active ? endpos + shift : rel_end), 63);
  }
  };
  if (sb <= kStageSubBits) phases(std::true_type{});
  else phases(std::false_type{});
  if (lane == 0) {
    l_off[kDecThreads] = base;
    l_off[kDecThreads + 3] = rounds;
    l_off[kDecThreads + 4] = 0;   // one record per lane (no boundaries inside the lanes' ranges)
    l_off[kDecThreads + 2] = 3;   // boundaries of the write pass's chain of groups (no fence: the consumer is a later kernel)
  }
}

// ---------------------------------------------------------------------------
// k_row_count_q: the records of a WIDE row (one that goes through k_row_window: wider than the
// LDS, e.g. the 16384-pixel rows of BASELINE config 4) -- FOUR per row-kernel lane, i.e. 4096
// sub-sequences per row, at the cost per step of k_row_count_w: the payload staged in LDS in
// blocks of 33 dwords and read on demand (15 VALU per step), where k_row_count<false> keeps a
// register window over global memory (28+, and 100 us per 1-Mbit row whatever the number of
// rows in flight: config-4 decode was bound by it).
// One workgroup per row.  Wavefront w takes sub-sequences [256 w, 256 w + 256) as four phases
// of 64 exactly like k_row_count_w takes a whole row (speculative lead-in, the chain inside a
// phase by DPP, a phase's first lane exact from the phase before); what is speculative here is
// where a WAVEFRONT's first sub-sequence starts.  The sixteen wavefronts then compare: wave w
// is right if it started where wave w - 1 ended (wave 0 is exact, the rest by induction);
// the lowest wave that is not walks its sub-sequences again from the exact position -- about
// one row in ten has such a wave (a 128-bit lead-in misses in < 1 % of the boundaries).
// Records: sub-sequence v is quarter v & 3 of lane v >> 2 in k_row_count<.., QTR>'s layout
// (lane_start / lane_off for quarter 0, lane_q for the others), flag 3 (boundaries of the
// write pass's chain of groups), and the index of the record that holds the first symbol of
// every 128 KiB window.
// ---------------------------------------------------------------------------
constexpr int kQPhases = 4;   // phases of 64 sub-sequences per wavefront: 16 x 4 x 64 = 4 x kDecThreads
// Where record q = 4 * lane + quarter lives (k_row_count<.., QTR>'s layout, k_row_window's rec_pos / rec_off).
__device__ __forceinline__ uint32_t *qrec_pos(uint32_t *l_start, uint32_t *l_q, uint32_t q) {
  const uint32_t t = q >> 2, k = q & 3u;
  return k == 0 ? l_start + t : l_q + (k - 1u) * kDecThreads + t;
}
__device__ __forceinline__ uint32_t *qrec_off(uint32_t *l_off, uint32_t *l_q, uint32_t q) {
  const uint32_t t = q >> 2, k = q & 3u;
  return k == 0 ? l_off + t : l_q + (2u + k) * kDecThreads + t;
}
__global__ __launch_bounds__(kDecThreads, 8) void k_row_count_q(Geom g, DecWs ws, const uint8_t *packed,
                                                               size_t in_stride, const uint32_t *sizes, int r0) {
  __shared__ __attribute__((aligned(16))) uint32_t gyx[2 * kTabEntries];
  __shared__ uint32_t nd[kMaxNodes + 1];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[kCountRowsW * kStageAlloc];
  __shared__ int s_flag;
  __shared__ uint32_t s_ws[kCountRowsW], s_we[kCountRowsW], s_wc[kCountRowsW], s_wl[kCountRowsW], s_bad;
  uint32_t *gy = gyx, *gx = gyx + kTabEntries;
  const int f = blockIdx.y, r = r0 + (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  DecFrame *df = ws.frames + f;
  if (tid == 0) {
    const int w = df->parse_status == 0 ? df->walk_status : 0;   // (k_row_count's rule)
    if (w && blockIdx.x == 0) atomicMax(&df->status, w);
    s_flag = df->status | w;
  }
  __syncthreads();
  const int failed = s_flag;
  const size_t ri = (size_t)f * g.rows + (size_t)r;
  uint32_t *l_start = ws.lane_start + ri * kDecThreads;
  uint32_t *l_off = ws.lane_off + ri * (kDecThreads + kRecHdr);
  uint32_t *l_q = ws.lane_q + ri * (6 * kDecThreads);
  if (tid == 0) l_off[kDecThreads + 2] = 0;                                   // not usable until proven otherwise
  if (tid >= kRecWin && tid < kRecHdr) l_off[kDecThreads + tid] = ~0u;        // no window index yet
  const uint32_t pay_off = ws.row_off[ri], pay_len = ws.row_len[ri];
  const unsigned long long rem64 = 8ull * pay_len;
  uint32_t sb1 = (uint32_t)((rem64 + kDecThreads - 1) / kDecThreads);   // the row kernels' sub-sequence: one chunk?
  sb1 = (sb1 + 31u) & ~31u;
  sb1 = sb1 < kMinSubBits ? kMinSubBits : sb1;
  constexpr uint32_t NSUB = 4u * (uint32_t)kDecThreads;
  uint32_t sb = (uint32_t)((rem64 + NSUB - 1u) / NSUB);
  sb = (sb + 31u) & ~31u;
  sb = sb < kMinSubBits ? kMinSubBits : sb;
  // Several chunks or nothing to do (k_row_count's rule), or sub-sequences beyond the staging
  // buffer (more than ~2.4 bits per symbol of a 16384-pixel row): the record stays unusable
  // and k_dec_huff decodes the row on its own.
  if (failed || sb1 > (uint32_t)g.max_sub || rem64 == 0 || g.row_block >= (1 << 22) || sb > kStageSubBits) return;
  {   // the write pass's step words next to the long-code descriptors (k_row_count_w)
    const uint32_t *nodes = ws.nodes + ((size_t)f * 2 + 1) * (kMaxNodes + 1);
    const int nn = df->s[1].num_nodes;
    for (int k = tid; k < nn; k += kDecThreads) nd[k] = nodes[k];
    const uint4 *gg = reinterpret_cast<const uint4 *>(ws.grp + ((size_t)f * 2 + 1) * (1u << kLutBits));
    for (int k = tid; k < (1 << kLutBits) / 2; k += kDecThreads) {
      const uint4 q = gg[k];
      reinterpret_cast<uint2 *>(gx)[k] = make_uint2(q.x, q.z);
      reinterpret_cast<uint2 *>(gy)[k] = make_uint2(q.y, q.w);
    }
    const uint4 *gs = reinterpret_cast<const uint4 *>(ws.sub + ((size_t)f * 2 + 1) * kSubEntries);
    for (int k = tid; k < kSubEntries / 2; k += kDecThreads) {
      const uint4 q = gs[k];
      reinterpret_cast<uint2 *>(gx + (1 << kLutBits))[k] = make_uint2(q.x, q.z);
      reinterpret_cast<uint2 *>(gy + (1 << kLutBits))[k] = make_uint2(q.y, q.w);
    }
  }
  __syncthreads();
  GrpTables tb;
  tb.grp = nullptr; tb.gx = gx; tb.gy = gy; tb.nd = nd;
  const uint32_t rem = (uint32_t)rem64;
  GReader rd;
  const uint32_t rel0 = rd.attach(packed + (size_t)f * in_stride, sizes[f], 8ull * pay_off);
  const uint32_t rel_end = rel0 + rem;
  const uint32_t lead = (uint32_t)g.lead_bits;
  uint32_t *stage = s_stage + wave * kStageAlloc;
  LdsBits bits;
  bits.base = lds_addr(stage);
  uint32_t w_start = rel_end, w_end = rel_end, w_last = rel_end, w_cnt = 0, rounds = 0;
  // All four phases of this wavefront.  first0 / exact0: where its first sub-sequence starts,
  // exactly -- or its nominal start, the real one to be found by a lead-in like any lane's.
  // The records go out as they are found, offsets relative to the wavefront's first symbol.
  auto run_wave = [&](uint32_t first0, bool exact0) {
    uint32_t first = first0, base = 0;
    bool have_start = false;
#pragma unroll 1
    for (int j = 0; j < kQPhases; ++j) {
      const uint32_t v = 256u * (uint32_t)wave + 64u * (uint32_t)j + (uint32_t)lane;
      const uint32_t nb0 = v ? rel0 + v * sb : rel0;              // nominal range of sub-sequence v
      uint32_t nlim = rel0 + (v + 1u) * sb;
      if (nlim > rel_end) nlim = rel_end;
      const bool active = nb0 < rel_end;
      const uint32_t pb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb0);
      if (pb0 >= rel_end) {   // a phase beyond the payload: its sub-sequences own nothing
        *qrec_pos(l_start, l_q, v) = rem;
        *qrec_off(l_off, l_q, v) = base;
        continue;
      }
      const bool spec0 = j == 0 && !exact0;   // the phase's first lane looks for its start as well
      uint32_t sfrom = pb0;
      if (spec0) sfrom = pb0 - rel0 > lead ? pb0 - lead : rel0;
      if (first < sfrom) sfrom = first;         // (an exact start in front of the nominal one: the group before ran long)
      const uint32_t w0 = sfrom >> 5;
      const uint32_t shift = 32u * w0;          // positions below: relative to the first staged dword
      wave_lds_sync();   // the walks of the phase before are done with the buffer
      for (uint32_t k = (uint32_t)lane; 4u * k < kStageWords; k += 64u) {
        const uint32_t w = w0 + 4u * k;
        uint4 x;
        if (w + 3u <= rd.jmax) {
          const PackedU4 u = *reinterpret_cast<const PackedU4 *>(rd.w + w);
          x.x = u.x; x.y = u.y; x.z = u.z; x.w = u.w;
        } else {
          x.x = rd.ld(w); x.y = rd.ld(w + 1u); x.z = rd.ld(w + 2u); x.w = rd.ld(w + 3u);
        }
        uint32_t *d = stage + 4u * k + (k >> 3);   // blocks of 33 (LdsBits)
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
        if ((k & 7u) == 0u && k) d[-1] = x.x;
      }
      wave_lds_sync();
      const uint32_t b0 = nb0 - shift, lim = nlim - shift, lo0 = rel0 - shift, fst = first - shift;
      uint32_t start = active ? b0 : rel_end - shift;
      if (lane == 0 && active && !spec0) start = fst;
      if ((lane > 0 || spec0) && active && lead) {   // (lo0 may lie below zero in this frame of reference: differences only)
        uint32_t from = start - lo0 > lead ? start - lead : lo0;
        if (from < sfrom - shift) from = sfrom - shift;   // (not in front of what is staged)
        uint32_t guess, none;
        grp_count_lds(bits, tb, from, start, &guess, &none);
        start = guess;
      }
      uint32_t endpos = start, cnt = 0;
      bool dirty = active;
      uint32_t T = (active ? b0 : rel_end - shift) + kJoinBits;
      if (T > lim || T < b0) T = lim;
      uint32_t posT = ~0u, cT = 0;
      for (;;) {
        if (dirty) {
          uint32_t p1, c1;
          grp_count_lds(bits, tb, start, T, &p1, &c1);
          if (p1 == posT) {
            cnt = c1 + (cnt - cT);
          } else {
            uint32_t c2;
            grp_count_lds(bits, tb, p1, lim, &endpos, &c2);
            cnt = c1 + c2;
          }
          posT = p1;
          cT = c1;
        }
        // The chain: a lane starts where its left neighbour ended; the first lane of a
        // speculative phase keeps the start its lead-in found.
        const uint32_t lane0 = spec0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)start) : fst;
        const uint32_t ns = wave_shr1_dpp(lane0, endpos);
        dirty = active && ns != start;
        if (active) start = ns;
        ++rounds;
        if (!__any(dirty ? 1 : 0)) break;
      }
      const uint32_t c = min(cnt, 0x3fffffu);
      const uint32_t incl = wave_scan_add_dpp(c);
      *qrec_pos(l_start, l_q, v) = active ? start + shift - rel0 : rem;
      *qrec_off(l_off, l_q, v) = base + incl - c;
      if (!have_start) { w_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)(start + shift)); have_start = true; }
      base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      first = (uint32_t)__builtin_amdgcn_readlane((int)(active ? endpos + shift : rel_end), 63);
      // where the chain of the row's LAST sub-sequence ends (the header's `endrel`)
      const int la = 63 - __clzll((long long)__ballot(active ? 1 : 0));
      w_last = (uint32_t)__builtin_amdgcn_readlane((int)(endpos + shift), la);
    }
    w_end = first;
    w_cnt = base;
    if (!have_start) w_start = rel_end;
  };
  const uint32_t nom0 = wave ? rel0 + 256u * (uint32_t)wave * sb : rel0;   // this wavefront's first nominal start
  const bool w_active = nom0 < rel_end;
  run_wave(nom0, wave == 0);
  // ---- the wavefronts agree on the chain ----
#pragma unroll 1
  for (int it = 0; it < kCountRowsW + 1; ++it) {
    if (lane == 0) { s_ws[wave] = w_start; s_we[wave] = w_end; s_wc[wave] = w_cnt; s_wl[wave] = w_last; }
    if (tid == 0) s_bad = 0xffffu;
    __syncthreads();
    // (a wavefront beyond the payload has nothing to be wrong about)
    const bool ok = wave == 0 || !w_active || s_ws[wave] == s_we[wave - 1];
    if (!ok && lane == 0) atomicMin(&s_bad, (uint32_t)wave);
    __syncthreads();
    const uint32_t bad = s_bad;
    if (bad == 0xffffu) break;
    if ((uint32_t)wave == bad) run_wave(s_we[wave - 1], true);   // everything in front of it is final: again, exactly
    __syncthreads();   // (s_ws / s_we / s_bad are rewritten at the top)
  }
  // ---- the offsets become the row's; the record that holds the first symbol of every window ----
  uint32_t wbase = 0, total = 0, endrel = 0;
  for (int w = 0; w < kCountRowsW; ++w) {
    const uint32_t c = s_wc[w];
    if (w < wave) wbase += c;
    total += c;
    if (w == 0 || rel0 + 256u * (uint32_t)w * sb < rel_end) endrel = s_wl[w] - rel0;
  }
  wave_lds_sync();   // (this wavefront's own stores above, before it reads them back)
  __threadfence_block();
  const uint32_t out_size = (uint32_t)g.row_block;
#pragma unroll 1
  for (int j = 0; j < kQPhases; ++j) {
    const uint32_t v = 256u * (uint32_t)wave + 64u * (uint32_t)j + (uint32_t)lane;
    uint32_t *po = qrec_off(l_off, l_q, v);
    // (read back what this wavefront stored above: device-scope loads, past the vector L1)
    const uint32_t rel = __hip_atomic_load(po, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the next record of this wavefront closes the range (its last one: the wavefront's total)
    const uint32_t nrel = (j == kQPhases - 1 && lane == 63)
                              ? w_cnt : __hip_atomic_load(qrec_off(l_off, l_q, v + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t off = wbase + rel, c = nrel - rel;
    *po = off;
    if (c) {
      for (uint32_t w = (off + kRowWindow - 1u) / kRowWindow; w * kRowWindow < off + c && w * kRowWindow < out_size &&
                                                                 w < (uint32_t)(kRecHdr - kRecWin); ++w)
        if (w) l_off[kDecThreads + kRecWin + w] = v;
    }
    wave_lds_sync();   // (lane l + 1 has read record v + 1's relative offset before lane l + 1 rewrites it -- see below)
  }
  if (tid == 0) {
    l_off[kDecThreads] = total;
    l_off[kDecThreads + 1] = endrel;
    l_off[kDecThreads + 3] = rounds;
    l_off[kDecThreads + 4] = 1;   // four records per lane
    l_off[kDecThreads + 2] = 3;   // boundaries of the write pass's chain of groups (no fence: the consumer is a later kernel)
  }
}

// ---------------------------------------------------------------------------
// k_row_window: write pass of rows whose symbols do not fit the LDS (wider than 4224
// pixels), one workgroup per 128 KiB WINDOW of a row's symbols (a channel plane of a
// 16384-pixel row).  Every lane looks at its own record from k_row_count and decodes its
// sub-sequence if its symbols touch the window -- about a quarter of the lanes of a
// 16384-pixel row, the two at the edges for both neighbours --, OR-ing the literals
// into the zeroed window in LDS exactly like the fused row kernel; the window then
// leaves as 16-byte stores.  (Storing every group straight to a pre-zeroed plane in HBM
// -- 64 different cache lines per wave instruction -- was 3.5 ms for a 16384 x 16384 frame
// against 1.0 ms for the same walk without the stores, plus 0.4 ms to clear the plane.)
// Rows without a usable fixpoint from k_row_count are k_dec_huff's, as before.
// ---------------------------------------------------------------------------
struct WinShared { int flag, err; unsigned long long endbit; };
__global__ __launch_bounds__(kDecThreads) void k_row_window(Geom g, DecWs ws, const uint8_t *packed,
                                                            size_t in_stride, const uint32_t *sizes, int r0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  LdsTables &T = *reinterpret_cast<LdsTables *>(smem);
  WinShared &sh = *reinterpret_cast<WinShared *>(smem + sizeof(LdsTables));
  uint8_t *win = smem + sizeof(LdsTables) + 64;          // kWinGuard + window + kWinGuard
  const int wi = blockIdx.x, r = r0 + (int)blockIdx.y, f = blockIdx.z, tid = threadIdx.x;
  DecFrame *df = ws.frames + f;
  const size_t ri = (size_t)f * g.rows + r;
  const uint32_t *pre_start = ws.lane_start + ri * kDecThreads, *pre_off = ws.lane_off + ri * (kDecThreads + kRecHdr);
  const uint32_t *lq = ws.lane_q ? ws.lane_q + ri * (6 * kDecThreads) : nullptr;
  // Everything from global memory in front of the first barrier.
  const uint32_t tot = pre_off[kDecThreads], usable = pre_off[kDecThreads + 2];
  const bool quarters = lq != nullptr && pre_off[kDecThreads + 4] != 0;   // four records per lane (k_row_count for wide rows)
  const uint32_t pay_off = ws.row_off[ri], pay_len = ws.row_len[ri], ssize = sizes[f];
  const unsigned long long P1 = 8ull * pay_len;
  // Record q of the row (quarters: 4 * lane + k, else the lane): where its tokens start (bits
  // from the row's first) and its first symbol; the record behind it closes its range.
  const uint32_t nrec = quarters ? 4u * (uint32_t)kDecThreads : (uint32_t)kDecThreads;
  auto rec_pos = [&](uint32_t q) -> uint32_t {
    if (q >= nrec) return (uint32_t)P1;
    if (!quarters) return pre_start[q];
    const uint32_t t = q >> 2, k = q & 3u;
    return k == 0 ? pre_start[t] : lq[(k - 1u) * kDecThreads + t];
  };
  auto rec_off = [&](uint32_t q) -> uint32_t {
    if (q >= nrec) return tot;
    if (!quarters) return pre_off[q];
    const uint32_t t = q >> 2, k = q & 3u;
    return k == 0 ? pre_off[t] : lq[(2u + k) * kDecThreads + t];
  };
  const uint32_t out_size = (uint32_t)g.row_block;
  const uint32_t w0 = (uint32_t)wi * kRowWindow, w1 = min(w0 + kRowWindow, out_size);
  // The window's first record, from the row's index.  (Checked below by the lane that gets
  // it: a record that does not hold the window's first symbol -- stale words -- sends the
  // workgroup through all records instead.)
  uint32_t q0 = 0;
  if (quarters && wi > 0 && wi < kRecHdr - kRecWin) {
    const uint32_t c = pre_off[kDecThreads + kRecWin + wi];
    if (c < nrec) q0 = c;
  }
  uint32_t q = q0 + (uint32_t)tid;
  uint32_t st_rel = rec_pos(q), nst_rel = rec_pos(q + 1u), off = rec_off(q), nxt = rec_off(q + 1u);
  if (tid == 0) { sh.flag = (df->status || usable == 0) ? 1 : 0; sh.err = 0; sh.endbit = ~0ull; }
  load_dec_tables(ws, df, f, 1, &T);
  const uint32_t span = (w1 - w0) + kWinGuard;           // guard + window bytes
  {
    uint4 z;
    z.x = z.y = z.z = z.w = 0;
    for (uint32_t k = tid; k < (span + kWinGuard + 15u) / 16u; k += kDecThreads) reinterpret_cast<uint4 *>(win)[k] = z;
  }
  const int stale = __syncthreads_or((tid == 0 && q0 != 0 && !(off <= w0 && w0 < nxt)) ? 1 : 0);
  if (sh.flag) return;   // frame failed, or the row is k_dec_huff's
  if (stale) {           // (uniform) the index did not name the window's first record: from the row's first
    q = (uint32_t)tid;
    st_rel = rec_pos(q); nst_rel = rec_pos(q + 1u); off = rec_off(q); nxt = rec_off(q + 1u);
  }
  const GrpTables tb = tables_of(&T);
  GReader rd;
  const uint32_t rel0 = rd.attach(packed + (size_t)f * in_stride, ssize, 8ull * pay_off);
  const uint32_t rel_out = out_size - w0 + kWinGuard;    // the block's end, same origin
  // Rounds of 1024 consecutive records (one round with one record per lane; with quarters
  // one per window unless the window's symbols take more than a quarter of the payload).
  for (;;) {
    if (q < nrec) {
      const uint32_t lim = nst_rel < (uint32_t)P1 ? rel0 + nst_rel : rel0 + (uint32_t)P1;
      const uint32_t cnt = nxt - off;
      // The record's symbols [off, off + cnt) against the window; positions are handed to the
      // write loops relative to the window's guard (they wrap below zero in front of it:
      // the clip test is one unsigned compare).
      const bool inside = off + cnt < out_size, exact = !inside && off < out_size;
      const bool touches = cnt > 0 && off < w1 && off + cnt > w0;
      if (touches || (exact && wi == (int)((out_size - 1u) / kRowWindow))) {
        uint32_t end_bp = ~0u;
        const uint32_t op = off - w0 + kWinGuard;
        if (inside) {
          if (!lean_write<true>(rd, tb, rel0 + st_rel, lim, op, win, span, usable == 3u)) sh.err = 1;
        } else if (exact) {
          if (!exact_write<true>(rd, tb, rel0 + st_rel, lim, op, rel_out, win, &end_bp, span)) sh.err = 1;
          if (end_bp != ~0u) sh.endbit = (unsigned long long)(end_bp - rel0);
        }
      }
    }
    // Another round while the records behind this one can still reach the window.
    const uint32_t qn = q - (uint32_t)tid + (uint32_t)kDecThreads;   // (uniform)
    const int more = __syncthreads_or((tid == kDecThreads - 1 && q < nrec && nxt < w1) ? 1 : 0);
    if (!more || qn >= nrec) break;
    q = qn + (uint32_t)tid;
    st_rel = rec_pos(q); nst_rel = rec_pos(q + 1u); off = rec_off(q); nxt = rec_off(q + 1u);
  }
  __syncthreads();
  // The window leaves as 16-byte stores (row blocks and windows are multiples of 16).
  uint8_t *dst = ws.fres_sym + (size_t)f * ws.fres_stride + (size_t)r * g.row_block + w0;
  for (uint32_t k = tid; k < (w1 - w0) / 16u; k += kDecThreads)
    reinterpret_cast<uint4 *>(dst)[k] = reinterpret_cast<const uint4 *>(win + kWinGuard)[k];
  // accept / reject like UncompressStream (huffman_dec.cpp:361-417): by the workgroup of
  // the row's last window, whose lanes include the one that completes the block.
  if (tid == 0 && w1 == out_size) {
    int bad = sh.err;
    if (tot < out_size) bad = 1;   // ran out of payload before the block was full
    const unsigned long long E = sh.endbit;
    if (!bad && !(E <= P1 && E + 8 > P1 && E > 0)) bad = 1;
    if (bad) atomicMax(&df->status, fmt_err(7, 1));
  } else if (tid == 0 && sh.err) {
    atomicMax(&df->status, fmt_err(7, 1));
  }
}

// k_dec_zero: the LRES symbol planes (16-byte units) and the two diagnostics arrays,
// one launch instead of three memsets in front of every decode.
__global__ __launch_bounds__(256) void k_dec_zero(uint4 *sym, uint32_t n16, uint32_t *a, uint32_t na, uint32_t *b,
                                                  uint32_t nb) {
  uint4 z;
  z.x = z.y = z.z = z.w = 0;
  const uint32_t k0 = blockIdx.x * (256u * 4u) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t k = k0 + (uint32_t)j * 256u;
    if (k < n16) sym[k] = z;
  }
  const uint32_t stride_t = gridDim.x * 256u, t0 = blockIdx.x * 256u + threadIdx.x;
  for (uint32_t k = t0; k < na; k += stride_t) a[k] = 0;
  for (uint32_t k = t0; k < nb; k += stride_t) b[k] = 0;
}

// k_dec_status: copy the per-frame verdict out of the workspace.
__global__ void k_dec_status(DecWs ws, int32_t *status, int batch) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < batch) {
    const DecFrame *df = ws.frames + f;
    const int w = df->parse_status == 0 ? df->walk_status : 0;   // (merged by k_row_count already unless no row was asked for)
    status[f] = df->status > w ? df->status : w;
  }
}

#define HIMG_LAUNCH(name, grid, block, ...)                    \
  do {                                                         \
    prof_begin(prof, #name, stream);                           \
    hipLaunchKernelGGL(name, grid, block, 0, stream, __VA_ARGS__); \
    prof_end(prof, stream);                                    \
  } while (0)

int loop_counts_read_dec(unsigned long long *out) { return loop_counts_read(out); }

bool dec_rows_fit_lds(const Geom &g) { return fused_layout(g.row_block).total <= 160u * 1024u; }

void launch_rowwalk_only(const Geom &g, const DecWs &ws, const uint8_t *d_packed, size_t in_stride,
                         const uint32_t *d_sizes, hipStream_t stream) {
  hipLaunchKernelGGL(k_dec_rowwalk, dim3(1), dim3(64), 0, stream, g, ws, d_packed, in_stride, d_sizes, 0x7fffffff, 0);
}

void launch_rowwalk_range(const Geom &g, const DecWs &ws, const uint8_t *d_packed, size_t in_stride,
                          const uint32_t *d_sizes, int row_end, bool resume, hipStream_t stream) {
  hipLaunchKernelGGL(k_dec_rowwalk, dim3(1), dim3(64), 0, stream, g, ws, d_packed, in_stride, d_sizes, row_end, resume ? 1 : 0);
}

void launch_decode(const Geom &g, const DecWs &ws, int batch, const uint8_t *d_packed,
                   size_t in_stride, const uint32_t *d_sizes, uint8_t *d_out,
                   int32_t *d_status, hipStream_t stream, Profiler *prof, bool allow_fused,
                   const DecStreams *ds, int r0, int r1,
                   const uint32_t *d_row_index, bool index_only, int phase) {
  // d_row_index: the FRES row index is given (k_dec_set_index instead of the serial
  // header walk; one frame).  index_only: container parse and row-header walk only --
  // the caller reads ws.row_off / ws.row_len / DecFrame::rows_first (rank 0 of a
  // row-sharded decode); with r1 == 0 the walk stops in front of the first row header
  // (rows_first alone: what a rank needs to send the head of the stream on its way).
  // phase: kDecHead = what needs only the head of the stream (container parse, LRES chain,
  // predictor inverse), kDecRows = the FRES rows (index, counts, row kernels, verdict);
  // a row-sharded decode launches them one by one, the rows once their bytes have arrived.
  constexpr int kWalkAll = 0x7fffffff;
  const bool do_head = (phase & kDecHead) != 0, do_rows = (phase & kDecRows) != 0;
  if (index_only) {
    HIMG_LAUNCH(k_dec_parse, dim3(batch), dim3(kParseThreads), g, ws, d_packed, in_stride, d_sizes);
    HIMG_LAUNCH(k_dec_rowwalk, dim3(batch), dim3(64), g, ws, d_packed, in_stride, d_sizes, r1 == 0 ? 0 : kWalkAll, 0);
    HIMG_LAUNCH(k_dec_status, dim3((batch + 63) / 64), dim3(64), ws, d_status, batch);
    return;
  }
  // Block rows [r0, r1) only (row-sharded decode: every rank decodes the small
  // LRES stream and walks all row headers, then its own FRES rows).
  const int nrows = r1 - r0;
  static const int rpc_env = [] { const char *e = getenv("HIMG_ROWS_PER_COUNT"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 0; }();
  // Rows per k_row_count workgroup (tuning knob): a workgroup loads the decode tables once
  // for its rows; a single frame has too few rows to fill the CUs that way.
  const long long all_rows = (long long)batch * nrows;
  const int rpc = rpc_env ? rpc_env : all_rows <= 512 ? 1 : all_rows <= 1024 ? 2 : kRowsPerCount;
  const unsigned gx = (unsigned)((((g.cols + 31) / 32) * 64 + 255) / 256);   // k_tile_inv: two lanes per tile, 32 tiles per wave
  // Fused row kernel when the row's symbols and the decode tables fit the 160 KiB
  // LDS (width <= 4352 for RGBA); the payload is read in place from L2.
  constexpr uint32_t kLdsMax = 160u * 1024u;
  const int wps = (allow_fused && fused_layout(g.row_block).total <= kLdsMax) ? 1 : 0;
  // A single large frame (nothing else to fill the GPU with while one lane walks its row
  // headers, 1.3 ms for 16384 x 16384): the rows in two ranges -- the walk of a range on
  // `side`, its counts on `side2`, its row kernels on the caller's stream, each behind the
  // stage before it and beside the next range's walk.  16384^2: 4.15 -> 3.32 ms, which is
  // the sum of the row kernels' own times (two ranges or four); a dependency across
  // streams costs tens of microseconds, so frames of fewer than 1024 block rows (4096^2:
  // 0.40 ms either way, 1024^2: 0.23 -> 0.30 ms) stay in one piece.
  const int seg_env = ds ? ds->walk_segs : 0;
  const int seg_want = seg_env ? (seg_env < 1 ? 1 : seg_env > kWalkSegs ? kWalkSegs : seg_env)
                               : (nrows >= 1024 ? (wps ? 2 : kWalkSegs) : 1);   // (rows through windows: three kernels per range in a pipeline)
  const int nseg = (ds && batch == 1 && !d_row_index && r0 == 0 && r1 == g.rows && nrows >= 16 * kWalkSegs) ? seg_want : 1;
  hipStream_t side = ds ? ds->side : nullptr;
  hipStream_t cnt_stream = !ds ? stream : nseg > 1 ? ds->side2 : ds->side;
  auto seg_lo = [&](int k) { return r0 + (int)((long long)nrows * k / nseg); };
  // Diagnostics and the LRES symbols (k_lres_write stores the non-zero ones only) are
  // cleared up front: k_row_count may start as soon as the parse and the walk are done.
  if (do_head) {
    const uint32_t n16 = (uint32_t)(((size_t)batch * ws.lres_stride + 15) / 16);   // (the stride is a multiple of 256)
    const uint32_t ns = (uint32_t)((size_t)batch * (g.rows + 1) * 8), nr = ws.rc_stats ? (uint32_t)((size_t)batch * g.rows * 8) : 0u;
    prof_begin(prof, "memset", stream);
    hipLaunchKernelGGL(k_dec_zero, dim3((n16 + 256 * 4 - 1) / (256 * 4)), dim3(256), 0, stream,
                       reinterpret_cast<uint4 *>(ws.lres_sym), n16, ws.stats, ns, ws.rc_stats, nr);
    prof_end(prof, stream);
  }
  // Fork: the serial FRES row-header walk runs on the side stream beside k_dec_parse.
  if (ds && do_rows) {
    (void)hipEventRecord(ds->ev_fork, stream);
    (void)hipStreamWaitEvent(side, ds->ev_fork, 0);
    if (d_row_index) {
      prof_begin(prof, "k_dec_set_index", side);
      hipLaunchKernelGGL(k_dec_set_index, dim3(1), dim3(256), 0, side, g, ws, d_row_index, d_sizes, r0, r1);
      prof_end(prof, side);
      (void)hipEventRecord(ds->ev_walk[0], side);
    } else {
      for (int k = 0; k < nseg; ++k) {
        prof_begin(prof, "k_dec_rowwalk", side);
        hipLaunchKernelGGL(k_dec_rowwalk, dim3(batch), dim3(64), 0, side, g, ws, d_packed, in_stride, d_sizes,
                           k + 1 < nseg ? seg_lo(k + 1) : kWalkAll, k > 0 ? 1 : 0);
        prof_end(prof, side);
        (void)hipEventRecord(ds->ev_walk[k], side);
      }
    }
  }
  if (do_head) HIMG_LAUNCH(k_dec_parse, dim3(batch), dim3(kParseThreads), g, ws, d_packed, in_stride, d_sizes);
  // The FRES counts need the decode tables (parse) and the row index (walk); their
  // stream joins this one again before the FRES row kernels.
  // (ds == nullptr: everything in line.)
  // Batches: one wavefront per row (k_row_count_w) -- from 8192 rows per call, below that
  // the 1024-lane kernel keeps the GPU busier (HIMG_COUNT_WAVE=0 / 1 forces either).
  const bool count_wave = g.count_wave >= 0 ? g.count_wave != 0 : all_rows >= 8192;   // HIMG_OPT_COUNT_WAVE
  auto row_count = [&](hipStream_t s, int a, int b) {
    if (b <= a) return;
    prof_begin(prof, "k_row_count", s);
    if (count_wave)
      hipLaunchKernelGGL(k_row_count_w, dim3((b - a + kCountRowsW - 1) / kCountRowsW, batch), dim3(kDecThreads), 0, s,
                         g, ws, d_packed, in_stride, d_sizes, a, b);
    else if (wps)
      hipLaunchKernelGGL(k_row_count<true>, dim3((b - a + rpc - 1) / rpc, batch), dim3(kDecThreads), 0, s, g, ws,
                         d_packed, in_stride, d_sizes, a, b, rpc);
    else if (ws.lane_q && g.wide_q && g.count_wave < 0 && (g.row_block % 16) == 0)
      // Rows that go through windows, at a bit rate the staged reader takes (the host's estimate
      // from the stream's size; a row beyond it is left to k_dec_huff): four records per lane
      // from 4096 sub-sequences.  HIMG_OPT_COUNT_WAVE = 0 / 1 force the other two forms.
      hipLaunchKernelGGL(k_row_count_q, dim3(b - a, batch), dim3(kDecThreads), 0, s, g, ws, d_packed, in_stride, d_sizes, a);
    else
      hipLaunchKernelGGL(k_row_count<false>, dim3((b - a + rpc - 1) / rpc, batch), dim3(kDecThreads), 0, s, g, ws,
                         d_packed, in_stride, d_sizes, a, b, rpc);
    prof_end(prof, s);
  };
  // The entropy pass of rows that do not fit the LDS, rows [a, b): every 128 KiB window of a
  // row's symbols is assembled in LDS (k_row_window) and stored whole; rows whose block is not a
  // multiple of 16 bytes (ragged widths of 1-3 channel frames) and rows without a usable
  // record take k_dec_huff's 32 KiB windows (which skips the others).
  const bool window = (g.row_block % 16) == 0;
  auto window_pass = [&](hipStream_t s, int a, int b) {
    if (b <= a) return;
    if (window) {
      const uint32_t lds = (uint32_t)sizeof(LdsTables) + 64u + kRowWindow + 2u * kWinGuard + 16u;
      const unsigned nwin = (unsigned)(((uint32_t)g.row_block + kRowWindow - 1u) / kRowWindow);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_row_window),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      prof_begin(prof, "k_row_window", s);
      hipLaunchKernelGGL(k_row_window, dim3(nwin, b - a, batch), dim3(kDecThreads), lds, s, g, ws, d_packed,
                         in_stride, d_sizes, a);
      prof_end(prof, s);
    }
    prof_begin(prof, "k_dec_huff", s);
    hipLaunchKernelGGL(k_dec_huff, dim3(b - a, batch), dim3(kDecThreads), 0, s, g, ws, d_packed, in_stride,
                       d_sizes, 1 + a, 1, window ? 2 : 1);
    prof_end(prof, s);
  };
  if (!do_rows) {
    // (nothing of the rows in this launch)
  } else if (ds) {
    (void)hipEventRecord(ds->ev_fork, stream);
    (void)hipStreamWaitEvent(cnt_stream, ds->ev_fork, 0);
    // They fill the CUs the latency-bound LRES kernels leave idle.
    for (int k = 0; k < nseg; ++k) {
      if (cnt_stream != side) (void)hipStreamWaitEvent(cnt_stream, ds->ev_walk[k], 0);
      row_count(cnt_stream, seg_lo(k), seg_lo(k + 1));
      // Rows that go through HBM: their entropy pass (k_row_window, k_dec_huff) needs the
      // stream's tables and the counts, NOT the low-res plane -- it follows the counts on
      // their stream, beside the LRES chain on the caller's, and only the transform waits for
      // both.  (Behind the LRES chain on the caller's stream these kernels were the critical
      // path of a single large frame: LRES chain 1.1 ms, then 1.8 ms of window / transform
      // launches one after the other -- 16384^2: 2.88 -> see DESIGN.md.)
      (void)hipEventRecord(ds->ev_cnt[k], cnt_stream);
      if (!wps) {
        // ... on a stream of its own where there are several row ranges: the counts of the next
        // range start when its walk is done, not behind this range's windows.
        hipStream_t ws_ = (nseg > 1 && ds->side3) ? ds->side3 : cnt_stream;
        if (ws_ != cnt_stream) (void)hipStreamWaitEvent(ws_, ds->ev_cnt[k], 0);
        window_pass(ws_, seg_lo(k), seg_lo(k + 1));
        (void)hipEventRecord(ds->ev_win[k], ws_);
      }
    }
  } else if (d_row_index) {
    HIMG_LAUNCH(k_dec_set_index, dim3(1), dim3(256), g, ws, d_row_index, d_sizes, r0, r1);
  } else {
    HIMG_LAUNCH(k_dec_rowwalk, dim3(batch), dim3(64), g, ws, d_packed, in_stride, d_sizes, kWalkAll, 0);
  }
  if (do_head) {
    // LRES: every chunk in parallel, chain verified, serial fallback if not.
    static const int lres_stage = getenv("HIMG_LRES_STAGE") ? atoi(getenv("HIMG_LRES_STAGE")) : 3;
    if (lres_stage & 1)
      HIMG_LAUNCH(k_lres_spec<true>, dim3(ws.lres_chunks, batch), dim3(kDecThreads), g, ws,
                  d_packed, in_stride, d_sizes);
    else
      HIMG_LAUNCH(k_lres_spec<false>, dim3(ws.lres_chunks, batch), dim3(kDecThreads), g, ws,
                  d_packed, in_stride, d_sizes);
    if (lres_stage & 2)
      HIMG_LAUNCH(k_lres_fix<true>, dim3(batch), dim3(kDecThreads), g, ws, d_packed, in_stride, d_sizes);
    else
      HIMG_LAUNCH(k_lres_fix<false>, dim3(batch), dim3(kDecThreads), g, ws, d_packed, in_stride, d_sizes);
    HIMG_LAUNCH(k_lres_write, dim3(ws.lres_chunks, batch), dim3(kDecThreads), g, ws, d_packed,
                in_stride, d_sizes);
    HIMG_LAUNCH(k_lres_finish, dim3((batch + 63) / 64), dim3(64), ws, batch);
    HIMG_LAUNCH(k_lres_unpredict, dim3((g.mcols + 4 * kUnpredWaves - 1) / (4 * kUnpredWaves), g.mrows, batch * g.C),
                dim3(64 * kUnpredWaves), g, ws);
  }
  if (!do_rows) return;
  if (wps) {
    // Rows per workgroup: as many as the transform has lanes for and the LDS holds
    // (4096-pixel rows: one).
    const int per_row = ((g.cols + 31) / 32) * 64;
    int rpw = 1;
    while (rpw < 8 && (rpw + 1) * per_row <= kDecThreads && fused_layout(g.row_block, rpw + 1).total <= kLdsMax) ++rpw;
    static const int rpw_env = getenv("HIMG_ROWS_PER_FUSED") ? atoi(getenv("HIMG_ROWS_PER_FUSED")) : 0;
    if (rpw_env > 0 && rpw_env < rpw) rpw = rpw_env;
    if (g.W == 4096 && g.C == 4 && (g.H & 7) == 0) rpw = 1;
    const uint32_t lds = fused_layout(g.row_block, rpw).total;
    // Persistent workgroups (HIMG_PERSIST_ROWS=0: one workgroup per grid element as before): as
    // many as run at once -- one per CU.
    static const int persist_env = getenv("HIMG_PERSIST_ROWS") ? atoi(getenv("HIMG_PERSIST_ROWS")) : 1;
    static const int n_cu = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      return n;
    }();
    // (Sixteen wavefronts of 120+ registers each: one workgroup per CU whatever the LDS leaves.)
    const int per_cu = 1;
#define HIMG_FUSED_LAUNCH(COLS, A, B)                                                           \
  do {                                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dec_row_fused<COLS>),           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
    const int gx_ = ((B) - (A) + rpw - 1) / rpw;                                                \
    const long long all_ = (long long)gx_ * batch;                                              \
    const int slots_ = persist_env > 1 ? persist_env : n_cu * per_cu;                           \
    const bool pers_ = persist_env != 0 && all_ > slots_;                                       \
    RowArgs ra_;                                                                                \
    ra_.g = g; ra_.ws = ws; ra_.packed = d_packed; ra_.in_stride = in_stride; ra_.sizes = d_sizes; \
    ra_.out_frames = d_out; ra_.r0 = (A); ra_.r1 = (B); ra_.rpw = rpw; ra_.gx = gx_; ra_.gy = batch; \
    ra_.next_dist = pers_ ? 0 : n_cu * per_cu;                                                 \
    hipLaunchKernelGGL((k_dec_row_fused<COLS>), dim3((unsigned)(pers_ ? slots_ : all_)),        \
                       dim3(kDecThreads), lds, stream, ra_);                                    \
  } while (0)
    for (int k = 0; k < nseg; ++k) {
      const int a = seg_lo(k), b = seg_lo(k + 1);
      if (ds) (void)hipStreamWaitEvent(stream, ds->ev_cnt[k], 0);
      else row_count(stream, a, b);
      if (b <= a) continue;
      prof_begin(prof, "k_dec_row_fused", stream);
      const bool whole4 = g.C == 4 && (g.W & 7) == 0 && (g.H & 7) == 0;   // FULL4
      static const bool no_cols = getenv("HIMG_NO_COLS") != nullptr;   // (A/B knob: the run-time-stride variant for every width but 4096)
      if (g.W == 4096 && whole4) HIMG_FUSED_LAUNCH(512, a, b);
      else if (g.W == 2048 && whole4 && !no_cols) HIMG_FUSED_LAUNCH(256, a, b);   // compile-time strides for the other
      else if (g.W == 1920 && whole4 && !no_cols) HIMG_FUSED_LAUNCH(240, a, b);   // BASELINE widths (config 3: 1920)
      else if (whole4) HIMG_FUSED_LAUNCH(-1, a, b);
      else HIMG_FUSED_LAUNCH(0, a, b);
      prof_end(prof, stream);
    }
#undef HIMG_FUSED_LAUNCH
  } else {
    // Symbols through HBM: the entropy pass (window_pass above; with side streams it ran on the
    // counts' stream already), then the transform, which needs the low-res plane as well.
    for (int k = 0; k < nseg; ++k) {
      const int a = seg_lo(k), b = seg_lo(k + 1);
      if (ds) (void)hipStreamWaitEvent(stream, ds->ev_win[k], 0);
      else { row_count(stream, a, b); window_pass(stream, a, b); }
      if (b <= a) continue;
      if (g.C == 4 && (g.W & 7) == 0 && (g.H & 7) == 0) HIMG_LAUNCH(k_tile_inv<true>, dim3(gx, b - a, batch), dim3(256), g, ws, d_out, a);
      else HIMG_LAUNCH(k_tile_inv<false>, dim3(gx, b - a, batch), dim3(256), g, ws, d_out, a);
    }
  }
  HIMG_LAUNCH(k_dec_status, dim3((batch + 63) / 64), dim3(64), ws, d_status, batch);
}

}  // namespace himg_dev
