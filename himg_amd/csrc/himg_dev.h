// himg_dev.h -- structures shared by the HIP kernels and the host orchestration.
//
// Vocabulary (follows the reference's domain):
//   block row  : one strip of 8 pixel rows = cols tiles = row_block symbols
//   LRES / FRES: the low-res and full-res payloads before entropy coding
//   span       : a contiguous symbol range entropy-coded by one workgroup
//                (FRES: one block row; LRES: kLresSpan symbols)
#ifndef HIMG_DEV_H_
#define HIMG_DEV_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace himg_dev {

constexpr int kNumSym = 261;      // huffman_common.h:18-20
constexpr int kHistStride = 264;  // padded row of a histogram / code table
constexpr int kLresSpan = 16384;  // LRES symbols per entropy span
constexpr int kIterSyms = 4096;   // symbols per workgroup iteration (256 thr x 16 B)
constexpr int kTreeStride = 384;  // bytes reserved per serialised tree (max 359)
constexpr int kMaxCodeLen = 32;   // reference keeps codes in uint32_t (huffman_enc.cpp:87)
constexpr int kHeadLen = 175;     // RIFF(12)+FRMT(19)+LMAP(136)+'LRES',size(8)

struct Geom {
  int W, H, C, stride;
  int rows, cols, mrows, mcols;
  int chan_size, lres_size;  // LRES payload per channel / total
  int row_block;             // cols*C*64: FRES symbols per block row
  int ycbcr;                 // effective flag: use_ycbcr && C >= 3
  int lres_spans;            // ceil(lres_size / kLresSpan)
  int use_blocks;            // FRES: block_size < in_size (huffman_enc.cpp:256)
  int fix_t2;                // decoder, opt-in: accept the encoder's own compressible streams (see himg_hip.h)
  int lres_serial;           // decoder test knob: distrust the parallel LRES chain, take the serial fallback
  int max_sub;               // decoder: longest sub-sequence in bits (4096; tests lower it to force several chunks per stream)
  int lead_bits;             // decoder: lead-in before a lane's nominal start (lean_fixpoint; 0 = start blind)
  // Kernel variants (context options, see himg_hip.h HIMG_OPT_*): -1 = chosen by the launch size.
  int prefetch_rows;         // decoder: the row kernel touches the packed bytes of the row that takes its CU next (HIMG_PREFETCH_ROWS=0 turns it off: an A/B knob)
  int count_wave;            // decoder: k_row_count_w (a wavefront per row) instead of k_row_count
  int wide_q;                // decoder, rows wider than the LDS: the host's estimate says a quarter sub-sequence of a row
                             // (1/4096 of its payload) fits k_row_count_q's staging buffer -- that kernel counts
  int emit_rows;             // encoder: k_emit_t<8> (a wavefront per row) instead of a workgroup per row
  int front;                 // encoder: k_front (averages + low-res plane + pixel stage in one pass over the pixels)
  int row_tokens;            // encoder: FRES rows as a token stream (k_tok -> k_emit_tok) instead of k_tok_hist / k_emit_t
                             // over the dense symbol plane twice
  long long frame_bytes;     // W*H*stride
  long long fres_size;       // rows*row_block
};

// Per-batch device workspace of the encoder. Every array is indexed
// [frame][...] with the stated per-frame stride (in elements).
struct EncWs {
  uint8_t *avg, *low;        size_t plane_stride;  // C*rows*cols (padded)
  uint8_t *lres_sym;         size_t lres_stride;
  uint8_t *fres_sym;         size_t fres_stride;
  uint32_t *hist;            // [f][2][kHistStride]      0 = LRES, 1 = FRES
  uint32_t *span_hist_l;     // [f][lres_spans][kHistStride]
  uint32_t *span_hist_f;     // [f][rows][kHistStride]
  uint32_t *lres_trail;      // [f][lres_spans]  trailing zeros | allzero<<31
  uint64_t *codes;           // [f][2][kHistStride]
  uint32_t *lens;            // [f][2][kHistStride]
  uint8_t *tree;             // [f][2][kTreeStride]
  uint32_t *tree_nbytes;     // [f][2]
  uint64_t *span_bit0;       // [f][lres_spans + rows] absolute start bit in the frame's output
  uint32_t *span_bits;       // [f][lres_spans + rows] payload bits of the span
  int32_t *status;           // [f]
  // Token stream of the FRES rows (k_tok -> k_emit_tok; nullptr: k_tok_hist / k_emit_t work on the dense symbol plane).
  // A block row is cut into tok_nseg segments of tok_seg symbols; segment s of row r of frame f owns
  // tok_cap 16-bit slots at tok + ((f * rows + r) * tok_nseg + s) * tok_cap and its slot count
  // (a multiple of 8 slots is always written: the tail is padded with no-op slots) in tok_cnt.
  uint16_t *tok;
  uint32_t *tok_cnt;         // [f][rows][tok_nseg]
  int tok_seg, tok_nseg, tok_cap;
};

// Token slots (16 bits; see k_tok): literal | zeros in front << 8 (literal 1..255, 0..255 zeros);
// 0 = no-op; a run of zeros on its own takes an even-aligned pair of slots: kTokRunMark, then its length.
constexpr uint32_t kTokRunMark = 0x0100u;
constexpr int kTokMaxRun = 255;          // zeros a literal slot can carry
constexpr int kTokIter = 2048;           // symbols per wavefront iteration of the tokeniser (32 per lane)
constexpr int kTokSegPad = 64;           // slots a segment can need beyond its symbol count (runs on their own, padding)

// Container bytes that do not depend on the pixel data, built on the host.
struct StaticChunks {
  uint8_t head[176];  // RIFF..HIMG FRMT LMAP 'LRES' <size>   (kHeadLen bytes)
  uint8_t mid[272];   // QCFG FMAP 'FRES' <size>
  int mid_len;        // 268 with chroma table, 236 without
};

struct ShiftTables {
  uint8_t s[2][64];  // [0] luma, [1] chroma; row-major coefficient position
};

struct LresTables {
  int16_t tab[128];   // low-res companding table (positive half)
  uint8_t code[512];  // code[delta + 255] for delta in [-255, 255]
};

// ---- decoder ---------------------------------------------------------------

constexpr int kLutBits = 11;  // width of the Huffman decode group table
constexpr int kDecThreads = 1024;
constexpr int kSubEntries = 1024;  // second-level decode table (codes longer than kLutBits)
constexpr int kSubMaxBits = 6;     // widest second-level sub-table
constexpr int kLresSubBits = 256;  // parallel LRES decode: payload bits per lane of a chunk (kDecThreads lanes)

struct DecStream {           // one Huffman stream (LRES or FRES) of one frame
  uint32_t payload_off;      // byte offset (in the packed stream) after the aligned tree
  uint32_t chunk_end;        // byte offset of the end of the chunk
  int32_t root;              // node index of the root
  int32_t num_nodes;
};

struct DecFrame {            // written by k_dec_parse, read by later kernels
  int32_t status;
  int32_t parse_status;      // k_dec_parse's own verdict (status collects the later kernels' as well)
  int32_t walk_status;       // k_dec_rowwalk's verdict: it runs beside k_dec_parse, k_row_count merges it
  uint32_t rows_first;       // k_dec_rowwalk: byte offset of the first FRES row header (0: not found)
  uint32_t walk_q, walk_r, walk_end;   // k_dec_rowwalk in several launches: where the next one resumes (walk_q 0: finished)
  int32_t ycbcr;
  DecStream s[2];
  int16_t lmap[128];         // decoder-side companding tables (positive halves)
  int16_t fmap[128];
  uint8_t shift[2][64];
  // The row kernels' small tables, the same for every block row of the frame -- built once by
  // k_dec_parse, copied (kRowTabWords / 4 x 16 bytes) by every row workgroup: the code byte ->
  // dequantised magnitude (int16 [256]), the shifts (u8 [2][64]), the shifts as packed pairs in
  // tile_plane's register order (u32 [2][32]) and the identity-test words (u32 [4]).
  alignas(16) uint32_t row_tabs[128 + 32 + 64 + 4];
};
constexpr int kRowTabWords = 128 + 32 + 64 + 4;
static_assert(kRowTabWords % 4 == 0, "whole 16-byte words");

constexpr int kLresMemoWords = 6;
// Header words behind a row's kDecThreads lane offsets (lane_off): [0] symbols of the row,
// [1] where the chain ends (bits), [2] valid flag, [3] diagnostics, [4] 1: lane_q holds
// three more boundaries inside every lane's range (wide rows), [8 + w] the QUARTER record
// (4 * lane + k) that holds the first symbol of window w of the row's symbols (w >= 1).
constexpr int kRecHdr = 72;
constexpr int kRecWin = 8;
struct DecWs {
  DecFrame *frames;          // [f]
  uint32_t *nodes;           // [f][2][522]  child a | child b << 10 | (symbol + 1) << 20 (1023: no child; 0: a branch)
  uint2 *grp;                // [f][2][1<<kLutBits] group table (kernels_dec.hip GrpTables)
  uint32_t *gyc;             // [f][2][1<<kLutBits] step words of the count-only groups (no four-byte limit)
  uint2 *sub;                // [f][2][kSubEntries] second-level entries for codes longer than kLutBits (.x byte / node, .y step word)
  uint32_t *row_off;         // [f][rows] payload byte offset of each FRES row
  uint32_t *row_len;         // [f][rows]
  uint8_t *lres_sym;         size_t lres_stride;
  uint8_t *fres_sym;         size_t fres_stride;
  uint8_t *low;              size_t plane_stride;
  // k_row_count -> k_dec_row_fused: per FRES row and lane the first owned token
  // (bits from the chunk start) and the exclusive prefix of the symbol counts.
  uint32_t *lane_start;      // [f][rows][kDecThreads]
  uint32_t *lane_off;        // [f][rows][kDecThreads + kRecHdr]: offsets, then total, chain end (bits), valid flag
                             // (1: the lanes' ranges are divided at token boundaries and a consumer finishes its
                             // range token by token; 3: k_row_count_w -- at boundaries of the WRITE pass's chain
                             // of groups: a consumer that walks those groups from its start lands on its limit)
  // Rows too wide for the LDS (k_row_window): three more boundaries inside every lane's range,
  // so that a 128 KiB window of the row's symbols has 1024 sub-sequences to walk, not 256.
  uint32_t *lane_q;          // [f][rows][6][kDecThreads]: positions of the boundaries at 1/4, 1/2, 3/4 (bits from the
                             // row's first), then the output offsets there; nullptr for rows the fused kernel takes
  uint32_t *parse_stats;     // [f][4] k_dec_parse phase cycles / 16
  uint32_t *stats;           // [f][rows+1][8] k_dec_huff counters (chunks, rounds, cycle splits)
  uint32_t *rc_stats;        // [f][rows][8] k_row_count phase cycles / 16 (slowest wave)
  // Parallel LRES decode (k_lres_spec / verify / write).
  int lres_chunks;           // chunk slots per frame (upper bound from lres_size)
  uint32_t *spec_start;      // [f][lres_chunks][1024] lane start, bits from the chunk's nominal start
  uint32_t *spec_endpos;     // [f][lres_chunks][1024] lane end, same origin
  uint32_t *spec_cnt;        // [f][lres_chunks][1024] symbols per lane
  uint32_t *spec_memo;       // [f][lres_chunks][1024][kLresMemoWords] starts the lane has decoded from, packed
  uint64_t *spec_end;        // [f][lres_chunks] payload bit where the chunk's speculative chain ends
  uint64_t *fix_end;         // [f][lres_chunks] the same after the correction pass
  uint64_t *spec_tot;        // [f][lres_chunks] symbols of the chunk
  uint64_t *ver_base;        // [f][lres_chunks] output offset of the chunk
  int32_t *ver_ok;           // [f] 1 when every chunk verified
  uint64_t *lres_endbit;     // [f] bits consumed when the LRES output became complete
};

// ---- launch wrappers (defined in kernels_enc.hip / kernels_dec.hip) --------

struct Profiler;  // host-side, see himg_hip.hip

// Does launch_encode tokenise the FRES rows into slots (and so need EncWs::tok) for this call?
bool enc_uses_row_tokens(const Geom &g, int batch);
// Symbols per token segment for this geometry (a multiple of kTokIter).
int enc_tok_seg(const Geom &g);
// Debug: expand frame `frame`'s token stream into symbols again (dst: fres_size bytes).
void launch_tok_expand(const Geom &g, const EncWs &ws, int frame, uint8_t *dst, hipStream_t stream);

void launch_encode(const Geom &g, const EncWs &ws, int batch, const uint8_t *d_frames,
                   uint8_t *d_out, size_t out_stride, uint32_t *d_sizes,
                   const StaticChunks &sc, const ShiftTables &st, const LresTables &lt,
                   const uint8_t *d_fmap_lut, hipStream_t stream, Profiler *prof,
                   hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join);

// The decoder's helper streams and events (owned by the context).
constexpr int kWalkSegs = 4;   // single large frames: at most this many row ranges whose walk / count / row kernels overlap
struct DecStreams {
  hipStream_t side = nullptr;    // the serial row-header walk; k_row_count behind it (batches)
  hipStream_t side2 = nullptr;   // single frames: k_row_count of a row range while `side` walks the next one
  hipStream_t side3 = nullptr;   // rows wider than the LDS: the entropy pass of a row range (k_row_window) while
                                 // `side2` counts the next one and the caller's stream runs the LRES chain / transforms
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_walk[kWalkSegs] = {}, ev_cnt[kWalkSegs] = {}, ev_win[kWalkSegs] = {};
  int walk_segs = 0;             // HIMG_WALK_SEGS at context creation (0: by frame size)
};

// The fused row kernel serves this geometry (a block row's symbols and the decode tables fit
// the LDS of one CU); otherwise the rows go through 128 KiB windows (k_row_window).
bool dec_rows_fit_lds(const Geom &g);
constexpr int kLoopCounters = 8;
// Loop trip counters of a -DHIMG_LOOP_COUNTS build (loop_counts.h): read and reset; -1 when not compiled in.
int loop_counts_read_enc(unsigned long long *out);
int loop_counts_read_dec(unsigned long long *out);

void launch_decode(const Geom &g, const DecWs &ws, int batch, const uint8_t *d_packed,
                   size_t in_stride, const uint32_t *d_sizes, uint8_t *d_out,
                   int32_t *d_status, hipStream_t stream, Profiler *prof, bool allow_fused,
                   const DecStreams *ds, int r0, int r1,
                   const uint32_t *d_row_index = nullptr, bool index_only = false, int phase = 3);
constexpr int kDecHead = 1, kDecRows = 2;   // launch_decode's phases
// The row-header walk of one frame alone (row-sharded decode: beside the head phase).
void launch_rowwalk_only(const Geom &g, const DecWs &ws, const uint8_t *d_packed, size_t in_stride,
                         const uint32_t *d_sizes, hipStream_t stream);
// ... up to (not including) block row `row_end`; resume: go on where the launch before stopped.
void launch_rowwalk_range(const Geom &g, const DecWs &ws, const uint8_t *d_packed, size_t in_stride,
                          const uint32_t *d_sizes, int row_end, bool resume, hipStream_t stream);

// Row-sharded encode of one frame (multi-GPU): phases between the collectives.
void launch_shard_stats(const Geom &g, const EncWs &ws, const uint8_t *d_frame_base,
                        const ShiftTables &st, const uint8_t *d_fmap_lut, int r0, int r1,
                        hipStream_t stream, Profiler *prof);
void launch_shard_row_bits(const Geom &g, const EncWs &ws, int r0, int r1, uint32_t *d_bits_out,
                           hipStream_t stream, Profiler *prof);
void launch_shard_emit(const Geom &g, const EncWs &ws, const StaticChunks &sc,
                       const uint32_t *d_all_row_bits, uint8_t *d_rel, size_t rel_cap,
                       uint32_t *d_rel_size, int r0, int r1, hipStream_t stream, Profiler *prof);
void launch_shard_assemble(const Geom &g, const EncWs &ws, const StaticChunks &sc,
                           const LresTables &lt, const uint32_t *d_all_row_bits,
                           const uint8_t *d_rel, size_t rel_bytes, uint8_t *d_out, size_t out_cap,
                           uint32_t *d_size, hipStream_t stream, Profiler *prof);

void launch_shard_head(const Geom &g, const EncWs &ws, const StaticChunks &sc, const LresTables &lt,
                       const uint32_t *d_all_row_bits, uint8_t *d_out, size_t out_cap, uint32_t *d_size,
                       uint32_t *d_head, int r0, int r1, hipStream_t stream, Profiler *prof);
void launch_shard_finish(const Geom &g, const EncWs &ws, uint8_t *d_out, size_t out_cap, const uint32_t *d_size,
                         hipStream_t stream, Profiler *prof);

// Stage timing hook: called before/after every kernel launch when profiling.
void prof_begin(Profiler *p, const char *stage, hipStream_t s);
void prof_end(Profiler *p, hipStream_t s);

}  // namespace himg_dev
#endif  // HIMG_DEV_H_
