// himg_hip.hip -- host side of the C ABI declared in include/himg_hip.h.
//
// Owns the device workspace, builds the data-independent container bytes and
// kernel tables on the host (they depend only on quality / geometry), and
// sequences the kernels of kernels_enc.hip / kernels_dec.hip.  There is no CPU
// fallback: every compute entry point needs a working HIP device.
#include "himg_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "himg_dev.h"
#include "himg_tables.h"

using namespace himg_dev;

namespace himg_dev {

struct Profiler {
  bool enabled = false;
  struct Rec { const char *name; hipEvent_t a, b; };
  std::vector<Rec> pending;
  struct Acc { std::string name; double ms = 0; int n = 0; };
  std::vector<Acc> acc;
  const char *cur = nullptr;
  hipEvent_t cur_a = nullptr;

  void collect() {
    for (auto &r : pending) {
      hipEventSynchronize(r.b);
      float ms = 0;
      hipEventElapsedTime(&ms, r.a, r.b);
      hipEventDestroy(r.a);
      hipEventDestroy(r.b);
      size_t i = 0;
      for (; i < acc.size(); ++i)
        if (acc[i].name == r.name) break;
      if (i == acc.size()) { acc.push_back(Acc()); acc.back().name = r.name; }
      acc[i].ms += ms;
      acc[i].n += 1;
    }
    pending.clear();
  }
};

void prof_begin(Profiler *p, const char *stage, hipStream_t s) {
  if (!p || !p->enabled) return;
  p->cur = stage;
  hipEventCreate(&p->cur_a);
  hipEventRecord(p->cur_a, s);
}
void prof_end(Profiler *p, hipStream_t s) {
  if (!p || !p->enabled) return;
  hipEvent_t b;
  hipEventCreate(&b);
  hipEventRecord(b, s);
  p->pending.push_back({p->cur, p->cur_a, b});
}

}  // namespace himg_dev

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  // Grow-only device allocation.
  bool reserve(size_t n) {
    if (n <= cap) return true;
    if (p) hipFree(p);
    p = nullptr;
    cap = 0;
    if (hipMalloc(&p, n) != hipSuccess) return false;
    cap = n;
    return true;
  }
  void release() {
    if (p) hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

struct himg_hip_ctx {
  int device = 0;
  std::string err;
  Profiler prof;
  hipStream_t last_stream = nullptr;
  // HIMG_FORCE_UNFUSED=1 routes every block row through the generic decode path
  // (symbols via HBM), the one rows wider than the LDS budget always take.
  bool allow_fused = true;
  bool use_side = true;  // HIMG_SIDE_STREAM=0 keeps the row-header walk on the caller's stream
  // Side stream + events: the decoder forks its serial row-header walk onto it.
  DecStreams dstr;
  // The encoder's side stream carries its LRES branch, which is on the critical path
  // (k_tree waits for it): default priority, its own events -- not the decoder's
  // lowest-priority stream, behind whose wide k_row_count launches of other contexts it
  // would queue.
  hipStream_t side_enc = nullptr;
  hipEvent_t ev_fork_e = nullptr, ev_join_e = nullptr;

  // Fixed table: LUT of the full-res companding search (FullResMapper is the
  // same for every quality, mapper.cpp:213-223), 32769 entries.
  DevBuf fmap_lut;
  size_t host_bytes = 0;   // bytes of the last host-API result still resident in h_out
  int fix_t2 = 0;          // HIMG_OPT_FIX_T2 (or HIMG_FIX_T2=1 in the environment)
  int max_sub = 4096;      // HIMG_MAX_SUB_BITS: test knob, see Geom::max_sub
  int lead_bits = 128;     // HIMG_LEAD_BITS: tuning knob, see Geom::lead_bits
  int lres_serial = 0;     // HIMG_FORCE_LRES_SERIAL=1: test knob, see Geom::lres_serial
  int count_wave = -1, emit_rows = -1;   // HIMG_OPT_COUNT_WAVE / _EMIT_ROWS (-1: by launch size)
  int row_tokens = -1;                   // HIMG_OPT_ROW_TOKENS (-1: by launch size)
  int front = -1;                        // HIMG_OPT_FRONT (-1: by launch size)
  // Batched host API: H2D of frame i+1, kernels of frame i and D2H of frame i-1 overlap
  // on three streams; staging is double buffered.
  struct Pipe {
    bool ready = false;
    hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    DevBuf in[2], out[2], meta[2];     // meta: u32 packed size, i32 status
    uint32_t *h_meta = nullptr;        // pinned mirror of meta[2] (+ the decode sizes), 2 x 4 words
  } pipe;

  // Encoder workspace.
  DevBuf e_planes, e_lres, e_fres, e_small, e_spanhist, e_tok, e_tokx;
  Geom enc_geom{};
  EncWs enc_ws{};
  int enc_batch = 0;
  bool enc_valid = false;

  // Decoder workspace.
  DevBuf d_frames, d_nodes, d_grp, d_gyc, d_sub, d_lane, d_rows, d_lres, d_fres, d_planes, d_sizes, d_stats, d_spec;
  Geom dec_geom{};
  DecWs dec_ws{};
  int dec_batch = 0;
  bool dec_valid = false;
  hipEvent_t ev_range[HIMG_MAX_WALK_RANGES] = {};   // himg_hip_decode_walk_ranges_device: behind every row range
  int n_ranges = 0;
  // What the last himg_hip_decode_head_device prepared (the frame tables and the low-res plane in
  // the decoder workspace): himg_hip_decode_rows_after_head_device must be handed the same
  // stream, geometry and HIP stream, with no other decode on this context in between (every
  // decode entry point goes through ensure_dec_ws, which drops the token).
  struct HeadToken {
    bool valid = false;
    int w = 0, h = 0, c = 0;
    const void *packed = nullptr;
    uint32_t size = 0;
    void *stream = nullptr;
  } head;

  // Staging for the host-buffer API.
  DevBuf h_in, h_out, h_sizes, h_status, h_index;
  uint32_t *hp_index = nullptr;          // pinned staging of the host row index (decode_core)
  size_t hp_index_cap = 0;               // in dwords

  // Packed sizes handed to the decoder: the caller's array (or a by-value argument)
  // may be gone before an asynchronous copy reads it, so the sizes are first copied
  // into a ctx-owned pinned slot; a slot is reused only after the copy that read it
  // has run (event).
  struct SizeRing {
    static constexpr int kSlots = 4;
    uint32_t *h[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[kSlots] = {0, 0, 0, 0};
    hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool busy[kSlots] = {false, false, false, false};
    int next = 0;
  } sizes_ring;

  // Row-sharded encode state (himg_hip_shard_*).
  struct {
    bool valid = false;
    Geom g{};
    StaticChunks sc{};
    ShiftTables st{};
    LresTables lt{};
    int r0 = 0, r1 = 0;
  } shard;
};

static int fail(himg_hip_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess) {
  if (ctx) {
    ctx->err = what;
    if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
  }
  return code;
}

#define HIP_TRY(ctx, expr)                                                  \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess) return fail(ctx, HIMG_ERR_HIP, #expr, e_);        \
  } while (0)

static bool make_geom(int width, int height, int pixel_stride, int num_channels, int use_ycbcr,
                      Geom *g) {
  if (width < 1 || height < 1 || num_channels < 1 || num_channels > 4 ||
      pixel_stride < num_channels)
    return false;
  g->W = width; g->H = height; g->C = num_channels; g->stride = pixel_stride;
  g->rows = (height + 7) >> 3; g->cols = (width + 7) >> 3;
  g->mrows = (g->rows + 15) / 16; g->mcols = (g->cols + 15) / 16;
  g->chan_size = g->mrows * g->mcols + g->rows * g->cols;   // downsampled.cpp:171-175
  const long long lres = (long long)g->chan_size * num_channels;
  const long long fres = (long long)g->rows * g->cols * 64 * num_channels;
  // The reference keeps every size in `int` (encoder.cpp:81,270,339-341).
  if (fres > 0x7fffffffLL || (long long)width * height * pixel_stride > 0x7fffffffLL) return false;
  g->lres_size = (int)lres;
  g->row_block = g->cols * num_channels * 64;
  g->ycbcr = (use_ycbcr && num_channels >= 3) ? 1 : 0;   // encoder.cpp:69
  g->lres_spans = (g->lres_size + kLresSpan - 1) / kLresSpan;
  g->use_blocks = g->rows > 1 ? 1 : 0;                   // block_size < in_size
  g->fix_t2 = 0;
  g->max_sub = 4096;
  g->lead_bits = 128;
  g->lres_serial = 0;
  g->count_wave = g->emit_rows = g->row_tokens = g->front = -1;
  g->wide_q = 0;
  { static const int pf = [] { const char *e = std::getenv("HIMG_PREFETCH_ROWS"); return e ? atoi(e) : 1; }(); g->prefetch_rows = pf; }
  g->frame_bytes = (long long)width * height * pixel_stride;
  g->fres_size = fres;
  return true;
}

extern "C" size_t himg_hip_max_packed_size(int width, int height, int num_channels) {
  Geom g;
  if (!make_geom(width, height, num_channels, num_channels, 1, &g)) return 0;
  // Payloads never exceed their symbol counts by more than the tree
  // (huffman_enc.cpp:242-244 assumes the same); plus row headers and chunks.
  size_t n = 12 + 19 + 136 + 8 + 72 + 188 + 8;
  n += (size_t)g.lres_size + kTreeStride;
  n += (size_t)g.fres_size + kTreeStride + 4u * (size_t)g.rows;
  return round_up(n + 64, 256);
}

extern "C" int himg_hip_create(int device, himg_hip_ctx **out) {
  if (!out) return HIMG_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return HIMG_ERR_HIP;
  if (device < 0 || device >= count) return HIMG_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return HIMG_ERR_HIP;
  himg_hip_ctx *ctx = new himg_hip_ctx();
  ctx->device = device;
  if (const char *e = std::getenv("HIMG_FORCE_UNFUSED")) ctx->allow_fused = !(e[0] == '1');
  if (const char *e = std::getenv("HIMG_WALK_SEGS")) ctx->dstr.walk_segs = atoi(e);
  if (const char *e = std::getenv("HIMG_SIDE_STREAM")) ctx->use_side = !(e[0] == '0');
  if (const char *e = std::getenv("HIMG_FIX_T2")) ctx->fix_t2 = e[0] == '1';
  if (const char *e = std::getenv("HIMG_FORCE_LRES_SERIAL")) ctx->lres_serial = e[0] == '1';
  if (const char *e = std::getenv("HIMG_MAX_SUB_BITS")) {
    const int v = std::atoi(e);
    if (v >= 128 && v <= 4096 && v % 32 == 0) ctx->max_sub = v;
  }
  if (const char *e = std::getenv("HIMG_COUNT_WAVE")) ctx->count_wave = atoi(e) ? 1 : 0;
  if (const char *e = std::getenv("HIMG_EMIT_ROWS")) ctx->emit_rows = atoi(e) ? 1 : 0;
  if (const char *e = std::getenv("HIMG_ROW_TOKENS")) ctx->row_tokens = atoi(e) ? 1 : 0;
  if (const char *e = std::getenv("HIMG_FRONT")) ctx->front = atoi(e) ? 1 : 0;
  if (const char *e = std::getenv("HIMG_LEAD_BITS")) {
    const int v = std::atoi(e);
    if (v >= 0 && v <= 4096) ctx->lead_bits = v;
  }
  // Companding LUT for every magnitude an int16 can take.
  std::vector<uint8_t> lut(32769);
  int16_t fmap[128];
  himg_tables_fullres_map(fmap);
  for (int a = 0; a <= 32767; ++a) lut[a] = himg_tables_map_to_8bit(fmap, a);
  lut[32768] = (uint8_t)(0u - himg_tables_map_to_8bit(fmap, -32768));  // |x| of -32768 (unreachable)
  if (!ctx->fmap_lut.reserve(round_up(lut.size(), 256)) ||
      hipMemcpy(ctx->fmap_lut.p, lut.data(), lut.size(), hipMemcpyHostToDevice) != hipSuccess) {
    delete ctx;
    return HIMG_ERR_HIP;
  }
  // The side stream carries the wide k_row_count next to the caller's stream's short
  // LRES kernels: lowest priority, so that those are placed first whenever a slot
  // frees up (they sit on the critical path of the frame, the counts do not).
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  bool ev_ok = true;
  for (int k = 0; k < kWalkSegs; ++k)
    ev_ok = ev_ok && hipEventCreateWithFlags(&ctx->dstr.ev_walk[k], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->dstr.ev_cnt[k], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->dstr.ev_win[k], hipEventDisableTiming) == hipSuccess;
  if (!ev_ok || hipStreamCreateWithPriority(&ctx->dstr.side, hipStreamNonBlocking, prio_least) != hipSuccess ||
      hipStreamCreateWithPriority(&ctx->dstr.side2, hipStreamNonBlocking, prio_least) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->dstr.ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->side_enc, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork_e, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join_e, hipEventDisableTiming) != hipSuccess) {
    himg_hip_destroy(ctx);
    return HIMG_ERR_HIP;
  }
  // The decoder's third helper stream IS the encoder's side stream (a context never encodes and
  // decodes at once): the runtime maps streams onto FOUR hardware queues (GPU_MAX_HW_QUEUES), and
  // a fifth stream of this context shared a queue with another one -- the encoder's LRES branch
  // then ran behind the pixel stage instead of beside it (+1.5 ms per 128-frame step, measured).
  ctx->dstr.side3 = ctx->side_enc;
  *out = ctx;
  return HIMG_OK;
}

extern "C" void himg_hip_destroy(himg_hip_ctx *ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  if (ctx->dstr.ev_fork) hipEventDestroy(ctx->dstr.ev_fork);
  for (int k = 0; k < kWalkSegs; ++k) {
    if (ctx->dstr.ev_walk[k]) hipEventDestroy(ctx->dstr.ev_walk[k]);
    if (ctx->dstr.ev_cnt[k]) hipEventDestroy(ctx->dstr.ev_cnt[k]);
    if (ctx->dstr.ev_win[k]) hipEventDestroy(ctx->dstr.ev_win[k]);
  }
  for (int k = 0; k < HIMG_MAX_WALK_RANGES; ++k)
    if (ctx->ev_range[k]) hipEventDestroy(ctx->ev_range[k]);
  if (ctx->dstr.side) hipStreamDestroy(ctx->dstr.side);
  if (ctx->dstr.side2) hipStreamDestroy(ctx->dstr.side2);
  if (ctx->ev_fork_e) hipEventDestroy(ctx->ev_fork_e);
  if (ctx->ev_join_e) hipEventDestroy(ctx->ev_join_e);
  if (ctx->side_enc) hipStreamDestroy(ctx->side_enc);
  for (int k = 0; k < himg_hip_ctx::SizeRing::kSlots; ++k) {
    if (ctx->sizes_ring.ev[k]) hipEventDestroy(ctx->sizes_ring.ev[k]);
    if (ctx->sizes_ring.h[k]) hipHostFree(ctx->sizes_ring.h[k]);
  }
  ctx->prof.collect();
  if (ctx->pipe.ready) {
    for (int k = 0; k < 2; ++k) {
      hipEventDestroy(ctx->pipe.ev_in[k]); hipEventDestroy(ctx->pipe.ev_k[k]); hipEventDestroy(ctx->pipe.ev_out[k]);
      ctx->pipe.in[k].release(); ctx->pipe.out[k].release(); ctx->pipe.meta[k].release();
    }
    hipStreamDestroy(ctx->pipe.s_in); hipStreamDestroy(ctx->pipe.s_comp); hipStreamDestroy(ctx->pipe.s_out);
    hipHostFree(ctx->pipe.h_meta);
  }
  DevBuf *all[] = {&ctx->fmap_lut, &ctx->e_planes, &ctx->e_lres, &ctx->e_fres, &ctx->e_small,
                   &ctx->e_spanhist, &ctx->e_tok, &ctx->e_tokx, &ctx->d_frames, &ctx->d_nodes, &ctx->d_grp, &ctx->d_gyc, &ctx->d_sub, &ctx->d_lane, &ctx->d_rows,
                   &ctx->d_lres, &ctx->d_fres, &ctx->d_planes, &ctx->d_sizes, &ctx->d_stats, &ctx->d_spec, &ctx->h_in,
                   &ctx->h_out, &ctx->h_sizes, &ctx->h_status, &ctx->h_index};
  for (DevBuf *b : all) b->release();
  if (ctx->hp_index) hipHostFree(ctx->hp_index);
  delete ctx;
}

extern "C" const char *himg_hip_last_error(const himg_hip_ctx *ctx) {
  return ctx ? ctx->err.c_str() : "no context";
}

extern "C" void himg_hip_free(void *p) { std::free(p); }

extern "C" void *himg_hip_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
extern "C" void himg_hip_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

extern "C" int himg_hip_get_option(himg_hip_ctx *ctx, int option, int *value) {
  if (!ctx || !value) return HIMG_ERR_ARG;
  if (option == HIMG_OPT_FIX_T2) { *value = ctx->fix_t2; return HIMG_OK; }
  if (option == HIMG_OPT_COUNT_WAVE) { *value = ctx->count_wave; return HIMG_OK; }
  if (option == HIMG_OPT_EMIT_ROWS) { *value = ctx->emit_rows; return HIMG_OK; }
  if (option == HIMG_OPT_ROW_TOKENS) { *value = ctx->row_tokens; return HIMG_OK; }
  if (option == HIMG_OPT_FRONT) { *value = ctx->front; return HIMG_OK; }
  return fail(ctx, HIMG_ERR_ARG, "unknown option");
}

extern "C" int himg_hip_set_option(himg_hip_ctx *ctx, int option, int value) {
  if (!ctx) return HIMG_ERR_ARG;
  if (option == HIMG_OPT_FIX_T2) { ctx->fix_t2 = value ? 1 : 0; return HIMG_OK; }
  const int tri = value < 0 ? -1 : (value ? 1 : 0);
  if (option == HIMG_OPT_COUNT_WAVE) { ctx->count_wave = tri; return HIMG_OK; }
  if (option == HIMG_OPT_EMIT_ROWS) { ctx->emit_rows = tri; return HIMG_OK; }
  if (option == HIMG_OPT_FRONT) { ctx->front = tri; return HIMG_OK; }
  if (option == HIMG_OPT_ROW_TOKENS) { ctx->row_tokens = value == 2 ? 2 : tri; return HIMG_OK; }   // (2: on + k_emit_tok's spelled-out path, a test knob)
  return fail(ctx, HIMG_ERR_ARG, "unknown option");
}

// ---------------------------------------------------------------------------
// Host-built tables and container bytes.
// ---------------------------------------------------------------------------
static void put_u32(uint8_t *p, uint32_t x) {
  p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); p[2] = (uint8_t)(x >> 16); p[3] = (uint8_t)(x >> 24);
}

static int build_static(const Geom &g, int quality, StaticChunks *sc, ShiftTables *st,
                        LresTables *lt) {
  memset(sc, 0, sizeof(*sc));
  uint8_t *h = sc->head;
  // encoder.cpp:111-129,139-166
  memcpy(h, "RIFF", 4); put_u32(h + 4, 0); memcpy(h + 8, "HIMG", 4);
  memcpy(h + 12, "FRMT", 4); put_u32(h + 16, 11);
  h[20] = 1; put_u32(h + 21, (uint32_t)g.W); put_u32(h + 25, (uint32_t)g.H);
  h[29] = (uint8_t)g.C; h[30] = (uint8_t)g.ycbcr;
  // encoder.cpp:88-89,168-184
  int16_t lmap[128], fmap[128];
  himg_tables_lowres_map(quality, lmap);
  memcpy(h + 31, "LMAP", 4);
  const int ln = himg_tables_mapping_function(lmap, h + 39);
  if (ln != 128) return HIMG_ERR_UNSUPPORTED;  // low-res tables never exceed 255
  put_u32(h + 35, (uint32_t)ln);
  memcpy(h + 167, "LRES", 4);  // size patched on the device
  // encoder.cpp:95-100,222-256
  himg_tables_shift(quality, 0, st->s[0]);
  himg_tables_shift(quality, 1, st->s[1]);
  uint8_t *m = sc->mid;
  int o = 0;
  memcpy(m + o, "QCFG", 4); put_u32(m + o + 4, g.ycbcr ? 64u : 32u); o += 8;
  for (int t = 0; t < (g.ycbcr ? 2 : 1); ++t)
    for (int i = 0; i < 32; ++i) m[o++] = (uint8_t)((st->s[t][2 * i] << 4) | st->s[t][2 * i + 1]);
  himg_tables_fullres_map(fmap);
  memcpy(m + o, "FMAP", 4);
  const int fn = himg_tables_mapping_function(fmap, m + o + 8);
  put_u32(m + o + 4, (uint32_t)fn);
  o += 8 + fn;
  memcpy(m + o, "FRES", 4); o += 8;  // size patched on the device
  sc->mid_len = o;
  // Low-res companding LUT over every possible prediction error.
  memcpy(lt->tab, lmap, sizeof(lmap));
  for (int d = -255; d <= 255; ++d) lt->code[d + 255] = himg_tables_map_to_8bit(lmap, d);
  lt->code[511] = 0;
  return HIMG_OK;
}

// ---------------------------------------------------------------------------
// Workspaces.
// ---------------------------------------------------------------------------
// Copy `n` packed sizes into the next pinned slot and start the H2D copy from there.
static int stage_sizes(himg_hip_ctx *ctx, const uint32_t *src, int n, hipStream_t s) {
  auto &r = ctx->sizes_ring;
  const int k = r.next;
  r.next = (k + 1) % himg_hip_ctx::SizeRing::kSlots;
  if (!r.ev[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&r.ev[k], hipEventDisableTiming));
  if (r.busy[k]) HIP_TRY(ctx, hipEventSynchronize(r.ev[k]));
  if (r.cap[k] < (size_t)n) {
    if (r.h[k]) hipHostFree(r.h[k]);
    r.h[k] = nullptr;
    r.cap[k] = 0;
    const size_t want = (size_t)n < 64 ? 64 : (size_t)n;
    HIP_TRY(ctx, hipHostMalloc((void **)&r.h[k], want * 4, hipHostMallocDefault));
    r.cap[k] = want;
  }
  memcpy(r.h[k], src, (size_t)n * 4);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_sizes.p, r.h[k], (size_t)n * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipEventRecord(r.ev[k], s));
  r.busy[k] = true;
  return HIMG_OK;
}

static int ensure_enc_ws(himg_hip_ctx *ctx, const Geom &g_in, int batch, bool allow_row_tokens = true, bool force_row_tokens = false) {
  EncWs &w = ctx->enc_ws;
  Geom g = g_in;
  g.row_tokens = ctx->row_tokens;
  // A row-sharded encode in progress belongs to the geometry it was started with.
  {
    const Geom &o = ctx->shard.g;
    if (ctx->shard.valid && (batch != 1 || o.W != g.W || o.H != g.H || o.C != g.C || o.stride != g.stride ||
                             o.ycbcr != g.ycbcr))
      ctx->shard.valid = false;
  }
  const size_t plane = round_up((size_t)g.C * g.rows * g.cols, 256);
  const size_t lres = round_up((size_t)g.lres_size + 16, 256);
  const size_t fres = round_up((size_t)g.fres_size + 16, 256);
  const int nsp = g.lres_spans + g.rows;
  // FRES rows as a token stream between the tokeniser and the bit packer (batches): 16-bit slots,
  // worst case 2 bytes per symbol (+ padding per segment), ~0.7 in use.
  const bool row_tok = allow_row_tokens && (force_row_tokens || himg_dev::enc_uses_row_tokens(g, batch));
  const int tok_seg = row_tok ? himg_dev::enc_tok_seg(g) : 0;
  const int tok_nseg = row_tok ? (g.row_block + tok_seg - 1) / tok_seg : 0;
  const int tok_cap = tok_seg + himg_dev::kTokSegPad;
  if (!ctx->e_planes.reserve(2 * plane * batch) || !ctx->e_lres.reserve(lres * batch) ||
      !ctx->e_fres.reserve(fres * batch) ||
      (row_tok && !ctx->e_tok.reserve((size_t)batch * g.rows * tok_nseg * ((size_t)tok_cap * 2 + 4))))
    return fail(ctx, HIMG_ERR_HIP, "encoder workspace allocation failed");
  // Small per-frame arrays, carved from one allocation.
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += round_up(bytes, 256); return o; };
  const size_t o_hist = carve((size_t)batch * 2 * kHistStride * 4);
  const size_t o_codes = carve((size_t)batch * 2 * kHistStride * 8);
  const size_t o_lens = carve((size_t)batch * 2 * kHistStride * 4);
  const size_t o_tree = carve((size_t)batch * 2 * kTreeStride);
  const size_t o_tnb = carve((size_t)batch * 2 * 4);
  const size_t o_trail = carve((size_t)batch * g.lres_spans * 4);
  const size_t o_bit0 = carve((size_t)batch * nsp * 8);
  const size_t o_bits = carve((size_t)batch * nsp * 4);
  const size_t o_status = carve((size_t)batch * 4);
  if (!ctx->e_small.reserve(off) ||
      !ctx->e_spanhist.reserve((size_t)batch * nsp * kHistStride * 4))
    return fail(ctx, HIMG_ERR_HIP, "encoder workspace allocation failed");
  uint8_t *sm = (uint8_t *)ctx->e_small.p;
  w.avg = (uint8_t *)ctx->e_planes.p;
  w.low = w.avg + plane * batch;
  w.plane_stride = plane;
  w.lres_sym = (uint8_t *)ctx->e_lres.p; w.lres_stride = lres;
  w.fres_sym = (uint8_t *)ctx->e_fres.p; w.fres_stride = fres;
  w.tok = row_tok ? (uint16_t *)ctx->e_tok.p : nullptr;
  w.tok_cnt = row_tok ? (uint32_t *)((uint8_t *)ctx->e_tok.p + (size_t)batch * g.rows * tok_nseg * (size_t)tok_cap * 2) : nullptr;
  w.tok_seg = tok_seg; w.tok_nseg = tok_nseg; w.tok_cap = tok_cap;
  w.hist = (uint32_t *)(sm + o_hist);
  w.codes = (uint64_t *)(sm + o_codes);
  w.lens = (uint32_t *)(sm + o_lens);
  w.tree = sm + o_tree;
  w.tree_nbytes = (uint32_t *)(sm + o_tnb);
  w.lres_trail = (uint32_t *)(sm + o_trail);
  w.span_bit0 = (uint64_t *)(sm + o_bit0);
  w.span_bits = (uint32_t *)(sm + o_bits);
  w.status = (int32_t *)(sm + o_status);
  w.span_hist_l = (uint32_t *)ctx->e_spanhist.p;
  w.span_hist_f = w.span_hist_l + (size_t)batch * g.lres_spans * kHistStride;
  ctx->enc_geom = g;
  ctx->enc_batch = batch;
  ctx->enc_valid = true;
  return HIMG_OK;
}

// Rows wider than the LDS: does a quarter sub-sequence (1 / 4096 of a row's payload) of the
// LARGEST stream of the call fit k_row_count_q's staging buffer, with a margin for rows above
// the frame's mean?  (An estimate from the stream's size; a row that exceeds it anyway is left
// to k_dec_huff by the kernel itself.)
static int wide_q_hint(const Geom &g, uint32_t max_packed_size) {
  if (himg_dev::dec_rows_fit_lds(g) || g.rows < 1) return 0;
  const double bits_per_quarter = 8.0 * (double)max_packed_size / ((double)g.rows * 4096.0);
  return bits_per_quarter <= 272.0 ? 1 : 0;   // kStageSubBits = 320: rows up to 17 % above the mean
}

static int ensure_dec_ws(himg_hip_ctx *ctx, const Geom &g, int batch) {
  DecWs &w = ctx->dec_ws;
  ctx->head.valid = false;   // whatever decode this is, it overwrites what a head phase left
  const size_t plane = round_up((size_t)g.C * g.rows * g.cols, 256);
  const size_t lres = round_up((size_t)g.lres_size + 16, 256);
  const size_t fres = round_up((size_t)g.fres_size + 16, 256);
  if (!ctx->d_frames.reserve(sizeof(DecFrame) * batch) ||
      !ctx->d_nodes.reserve((size_t)batch * 2 * (2 * kNumSym) * 4) ||
      !ctx->d_grp.reserve((size_t)batch * 2 * (1u << kLutBits) * 8) ||
      !ctx->d_gyc.reserve((size_t)batch * 2 * (1u << kLutBits) * 4) ||
      !ctx->d_sub.reserve((size_t)batch * 2 * kSubEntries * 8) ||
      !ctx->d_lane.reserve((size_t)batch * g.rows * (2 * kDecThreads + himg_dev::kRecHdr + (himg_dev::dec_rows_fit_lds(g) ? 0 : 6 * kDecThreads)) * 4) ||
      !ctx->d_rows.reserve((size_t)batch * g.rows * 4 * 2) || !ctx->d_lres.reserve(lres * batch) ||
      !ctx->d_fres.reserve(fres * batch) || !ctx->d_planes.reserve(plane * batch) ||
      !ctx->d_sizes.reserve((size_t)batch * 4) ||
      !ctx->d_stats.reserve(((size_t)batch * (g.rows + 1) * 8 + (size_t)batch * 4 + (size_t)batch * g.rows * 8) * 4))
    return fail(ctx, HIMG_ERR_HIP, "decoder workspace allocation failed");
  w.frames = (DecFrame *)ctx->d_frames.p;
  w.nodes = (uint32_t *)ctx->d_nodes.p;
  w.grp = (uint2 *)ctx->d_grp.p;
  w.gyc = (uint32_t *)ctx->d_gyc.p;
  w.sub = (uint2 *)ctx->d_sub.p;
  w.lane_start = (uint32_t *)ctx->d_lane.p;
  w.lane_off = w.lane_start + (size_t)batch * g.rows * kDecThreads;
  w.lane_q = himg_dev::dec_rows_fit_lds(g) ? nullptr : w.lane_off + (size_t)batch * g.rows * (kDecThreads + himg_dev::kRecHdr);
  w.row_off = (uint32_t *)ctx->d_rows.p;
  w.row_len = w.row_off + (size_t)batch * g.rows;
  w.lres_sym = (uint8_t *)ctx->d_lres.p; w.lres_stride = lres;
  w.fres_sym = (uint8_t *)ctx->d_fres.p; w.fres_stride = fres;
  w.low = (uint8_t *)ctx->d_planes.p; w.plane_stride = plane;
  w.stats = (uint32_t *)ctx->d_stats.p;
  w.parse_stats = w.stats + (size_t)batch * (g.rows + 1) * 8;
  w.rc_stats = w.parse_stats + (size_t)batch * 4;
  {
    // LRES payload <= lres_size + tree bytes (huffman_enc.cpp:242-244).
    const size_t max_bits = 8ull * ((size_t)g.lres_size + kTreeStride);
    const size_t cb = (size_t)kDecThreads * kLresSubBits;   // bits per chunk (kLresChunkBits of kernels_dec.hip)
    int nch = (int)((max_bits + cb - 1) / cb);
    if (nch > 1024) nch = 1024;  // beyond this the serial path takes over (k_lres_fix)
    w.lres_chunks = nch;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += round_up(bytes, 256); return o; };
    const size_t o_ss = carve((size_t)batch * nch * kDecThreads * 4);
    const size_t o_sc = carve((size_t)batch * nch * kDecThreads * 4);
    const size_t o_se = carve((size_t)batch * nch * 8);
    const size_t o_st = carve((size_t)batch * nch * 8);
    const size_t o_sp = carve((size_t)batch * nch * kDecThreads * 4);
    const size_t o_fe = carve((size_t)batch * nch * 8);
    const size_t o_vb = carve((size_t)batch * nch * 8);
    const size_t o_ok = carve((size_t)batch * 4);
    const size_t o_eb = carve((size_t)batch * 8);
    const size_t o_mm = carve((size_t)batch * nch * kDecThreads * 4 * himg_dev::kLresMemoWords);
    if (!ctx->d_spec.reserve(off)) return fail(ctx, HIMG_ERR_HIP, "decoder workspace allocation failed");
    uint8_t *b = (uint8_t *)ctx->d_spec.p;
    w.spec_start = (uint32_t *)(b + o_ss); w.spec_cnt = (uint32_t *)(b + o_sc);
    w.spec_end = (uint64_t *)(b + o_se); w.spec_tot = (uint64_t *)(b + o_st);
    w.spec_endpos = (uint32_t *)(b + o_sp); w.fix_end = (uint64_t *)(b + o_fe);
    w.ver_base = (uint64_t *)(b + o_vb); w.ver_ok = (int32_t *)(b + o_ok);
    w.lres_endbit = (uint64_t *)(b + o_eb);
    w.spec_memo = (uint32_t *)(b + o_mm);
  }
  ctx->dec_geom = g;
  ctx->dec_batch = batch;
  ctx->dec_valid = true;
  return HIMG_OK;
}

// Text the reference prints for a failed stage (decoder.cpp:96-135,232,287,345).
static std::string format_message(int32_t st) {
  static const char *kStage[] = {"", "Not a RIFF HIMG file.\n", "Error decoding header.\n",
                                 "Error decoding low-res mapping function.\n",
                                 "Error decoding low-res data.\n",
                                 "Error decoding quantization configuration.\n",
                                 "Error decoding full-res mapping function.\n",
                                 "Error decoding full-res data.\n"};
  std::string m;
  if (st & 0x100) m += "Error: Invalid Huffman data.\n";
  const int stage = (st >> 4) & 7;
  m += kStage[stage];
  return m;
}

static int status_to_code(int32_t st) {
  st &= 15;
  switch (st) {
    case 0: return HIMG_OK;
    case 1: return HIMG_ERR_ARG;
    case 3: return HIMG_ERR_UNSUPPORTED;
    case 4: return HIMG_ERR_FORMAT;
    case 5: return HIMG_ERR_CAPACITY;
    default: return HIMG_ERR_HIP;
  }
}

// ---------------------------------------------------------------------------
// Device-resident API.
// ---------------------------------------------------------------------------
__global__ void k_copy_status(const int32_t *src, int32_t *dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

extern "C" int himg_hip_encode_device(himg_hip_ctx *ctx, const void *d_frames, int batch, int width,
                                      int height, int pixel_stride, int num_channels, int quality,
                                      int use_ycbcr, void *d_out, size_t out_stride,
                                      uint32_t *d_sizes, int32_t *d_status, void *stream) {
  if (!ctx || !d_frames || !d_out || !d_sizes || batch < 1 || batch > 65535) return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, pixel_stride, num_channels, use_ycbcr, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  if (g.rows > 65535 || batch * g.C > 65535) return fail(ctx, HIMG_ERR_UNSUPPORTED, "grid too large");
  if ((out_stride & 255) || out_stride < 1024 || ((uintptr_t)d_out & 15) || ((uintptr_t)d_frames & 15))
    return fail(ctx, HIMG_ERR_ARG, "out_stride must be a multiple of 256; buffers 16-byte aligned");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_enc_ws(ctx, g, batch);
  if (rc) return rc;
  StaticChunks sc;
  ShiftTables st;
  LresTables lt;
  rc = build_static(g, quality, &sc, &st, &lt);
  if (rc) return fail(ctx, rc, "unsupported table configuration");
  hipStream_t s = (hipStream_t)stream;
  ctx->last_stream = s;
  g.emit_rows = ctx->emit_rows;
  g.row_tokens = ctx->row_tokens;
  g.front = ctx->front;
  launch_encode(g, ctx->enc_ws, batch, (const uint8_t *)d_frames, (uint8_t *)d_out, out_stride,
                d_sizes, sc, st, lt, (const uint8_t *)ctx->fmap_lut.p, s, &ctx->prof,
                ctx->use_side ? ctx->side_enc : nullptr, ctx->ev_fork_e, ctx->ev_join_e);
  if (d_status)
    hipLaunchKernelGGL(k_copy_status, dim3((batch + 63) / 64), dim3(64), 0, s, ctx->enc_ws.status,
                       d_status, batch);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_decode_device(himg_hip_ctx *ctx, const void *d_packed, size_t in_stride,
                                      const uint32_t *h_sizes, int batch, int width, int height,
                                      int num_channels, void *d_out, int32_t *d_status,
                                      void *stream) {
  if (!ctx || !d_packed || !h_sizes || !d_out || !d_status || batch < 1 || batch > 65535)
    return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, num_channels, num_channels, 1, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  g.fix_t2 = ctx->fix_t2;
  g.max_sub = ctx->max_sub;
  g.lead_bits = ctx->lead_bits;
  g.lres_serial = ctx->lres_serial;
  g.count_wave = ctx->count_wave;
  { uint32_t mx = 0; for (int i = 0; i < batch; ++i) mx = h_sizes[i] > mx ? h_sizes[i] : mx; g.wide_q = wide_q_hint(g, mx); }
  if (g.rows + 1 > 65535 || batch * g.C > 65535) return fail(ctx, HIMG_ERR_UNSUPPORTED, "grid too large");
  if ((in_stride & 3) || ((uintptr_t)d_packed & 15) || ((uintptr_t)d_out & 15))
    return fail(ctx, HIMG_ERR_ARG, "in_stride must be a multiple of 4; buffers 16-byte aligned");
  for (int i = 0; i < batch; ++i)
    if (((size_t)h_sizes[i] + 3) / 4 * 4 > in_stride)
      return fail(ctx, HIMG_ERR_ARG, "in_stride must cover every stream rounded up to 4 bytes");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dec_ws(ctx, g, batch);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  ctx->last_stream = s;
  rc = stage_sizes(ctx, h_sizes, batch, s);
  if (rc) return rc;
  launch_decode(g, ctx->dec_ws, batch, (const uint8_t *)d_packed, in_stride,
                (const uint32_t *)ctx->d_sizes.p, (uint8_t *)d_out, d_status, s, &ctx->prof,
                ctx->allow_fused, ctx->use_side ? &ctx->dstr : nullptr, 0,
                g.rows);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_decode_rows_device(himg_hip_ctx *ctx, const void *d_packed, uint32_t packed_size,
                                           int width, int height, int num_channels, int row0,
                                           int row1, void *d_out_rows, int32_t *d_status,
                                           void *stream) {
  if (!ctx || !d_packed || !d_out_rows || !d_status) return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, num_channels, num_channels, 1, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  g.fix_t2 = ctx->fix_t2;
  g.max_sub = ctx->max_sub;
  g.lead_bits = ctx->lead_bits;
  g.lres_serial = ctx->lres_serial;
  g.count_wave = ctx->count_wave;
  g.wide_q = wide_q_hint(g, packed_size);
  if (row0 < 0 || row1 < row0 || row1 > g.rows) return fail(ctx, HIMG_ERR_ARG, "bad row range");
  if (g.rows + 1 > 65535 || g.C > 65535) return fail(ctx, HIMG_ERR_UNSUPPORTED, "grid too large");
  if (((uintptr_t)d_packed & 15) || ((uintptr_t)d_out_rows & 15))
    return fail(ctx, HIMG_ERR_ARG, "buffers must be 16-byte aligned");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dec_ws(ctx, g, 1);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  ctx->last_stream = s;
  rc = stage_sizes(ctx, &packed_size, 1, s);
  if (rc) return rc;
  // The kernels address pixel rows of the whole frame; hand them a virtual frame
  // base so that block row row0 lands at the start of d_out_rows.
  uint8_t *base = (uint8_t *)d_out_rows - (size_t)8 * row0 * g.W * g.C;
  launch_decode(g, ctx->dec_ws, 1, (const uint8_t *)d_packed, ((size_t)packed_size + 3) / 4 * 4,
                (const uint32_t *)ctx->d_sizes.p, base, d_status, s, &ctx->prof, ctx->allow_fused,
                ctx->use_side ? &ctx->dstr : nullptr, row0, row1);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

// Row index of one stream in HBM (rank 0 of a row-sharded decode): container parse and
// the serial row-header walk only.  d_row_index: [rows] payload offsets then [rows]
// lengths; d_rows_first: offset of the first row header (everything in front of it --
// container chunks, LRES stream, FRES tree -- is what every rank needs).
extern "C" int himg_hip_decode_index_device(himg_hip_ctx *ctx, const void *d_packed, uint32_t packed_size,
                                            int width, int height, int num_channels,
                                            uint32_t *d_row_index, uint32_t *d_rows_first,
                                            int32_t *d_status, void *stream) {
  if (!ctx || !d_packed || !d_row_index || !d_rows_first || !d_status) return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, num_channels, num_channels, 1, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  g.fix_t2 = ctx->fix_t2;
  if (g.rows + 1 > 65535) return fail(ctx, HIMG_ERR_UNSUPPORTED, "grid too large");
  if ((uintptr_t)d_packed & 15) return fail(ctx, HIMG_ERR_ARG, "buffers must be 16-byte aligned");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dec_ws(ctx, g, 1);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  ctx->last_stream = s;
  rc = stage_sizes(ctx, &packed_size, 1, s);
  if (rc) return rc;
  launch_decode(g, ctx->dec_ws, 1, (const uint8_t *)d_packed, ((size_t)packed_size + 3) / 4 * 4,
                (const uint32_t *)ctx->d_sizes.p, nullptr, d_status, s, &ctx->prof, ctx->allow_fused,
                nullptr, 0, g.rows, nullptr, true);
  HIP_TRY(ctx, hipMemcpyAsync(d_row_index, ctx->dec_ws.row_off, (size_t)g.rows * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(d_row_index + g.rows, ctx->dec_ws.row_len, (size_t)g.rows * 4,
                              hipMemcpyDeviceToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(d_rows_first, &ctx->dec_ws.frames[0].rows_first, 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

// Block rows [row0, row1) with the row index supplied (no header walk: the buffer
// only has to hold the bytes in front of the first row header and the payloads of
// these rows, each at its offset in the stream).  phase: kDecHead | kDecRows.
static int decode_rows_indexed(himg_hip_ctx *ctx, const void *d_packed, uint32_t packed_size, int width,
                               int height, int num_channels, int row0, int row1, const uint32_t *d_row_index,
                               void *d_out_rows, int32_t *d_status, void *stream, int phase) {
  Geom g;
  if (!make_geom(width, height, num_channels, num_channels, 1, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  g.fix_t2 = ctx->fix_t2;
  g.max_sub = ctx->max_sub;
  g.lead_bits = ctx->lead_bits;
  g.lres_serial = ctx->lres_serial;
  g.count_wave = ctx->count_wave;
  g.wide_q = wide_q_hint(g, packed_size);
  if (row0 < 0 || row1 < row0 || row1 > g.rows) return fail(ctx, HIMG_ERR_ARG, "bad row range");
  if (g.rows + 1 > 65535 || g.C > 65535) return fail(ctx, HIMG_ERR_UNSUPPORTED, "grid too large");
  if (((uintptr_t)d_packed & 15) || ((uintptr_t)d_out_rows & 15))
    return fail(ctx, HIMG_ERR_ARG, "buffers must be 16-byte aligned");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dec_ws(ctx, g, 1);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  ctx->last_stream = s;
  rc = stage_sizes(ctx, &packed_size, 1, s);
  if (rc) return rc;
  uint8_t *base = d_out_rows ? (uint8_t *)d_out_rows - (size_t)8 * row0 * g.W * g.C : nullptr;
  launch_decode(g, ctx->dec_ws, 1, (const uint8_t *)d_packed, ((size_t)packed_size + 3) / 4 * 4,
                (const uint32_t *)ctx->d_sizes.p, base, d_status, s, &ctx->prof, ctx->allow_fused,
                ctx->use_side ? &ctx->dstr : nullptr, row0, row1, d_row_index, false, phase);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_decode_rows_indexed_device(himg_hip_ctx *ctx, const void *d_packed,
                                                   uint32_t packed_size, int width, int height,
                                                   int num_channels, int row0, int row1,
                                                   const uint32_t *d_row_index, void *d_out_rows,
                                                   int32_t *d_status, void *stream) {
  if (!ctx || !d_packed || !d_out_rows || !d_status || !d_row_index) return HIMG_ERR_ARG;
  return decode_rows_indexed(ctx, d_packed, packed_size, width, height, num_channels, row0, row1, d_row_index,
                             d_out_rows, d_status, stream, himg_dev::kDecHead | himg_dev::kDecRows);
}

// The same in two launches, for a rank whose rows' bytes arrive later than the head of
// the stream: decode_head_device needs only the bytes in front of the first row header
// (container parse, LRES chain, predictor inverse: the low-res plane of the whole frame),
// decode_rows_after_head_device -- same context, same stream, same geometry -- the row
// index and the rows' bytes.
extern "C" int himg_hip_decode_head_device(himg_hip_ctx *ctx, const void *d_packed, uint32_t packed_size,
                                           int width, int height, int num_channels, void *stream) {
  if (!ctx || !d_packed) return HIMG_ERR_ARG;
  const int rc = decode_rows_indexed(ctx, d_packed, packed_size, width, height, num_channels, 0, 0, nullptr, nullptr,
                                     nullptr, stream, himg_dev::kDecHead);
  if (rc == HIMG_OK) {
    ctx->head.valid = true;
    ctx->head.w = width; ctx->head.h = height; ctx->head.c = num_channels;
    ctx->head.packed = d_packed; ctx->head.size = packed_size; ctx->head.stream = stream;
  }
  return rc;
}

extern "C" int himg_hip_decode_rows_after_head_device(himg_hip_ctx *ctx, const void *d_packed,
                                                      uint32_t packed_size, int width, int height,
                                                      int num_channels, int row0, int row1,
                                                      const uint32_t *d_row_index, void *d_out_rows,
                                                      int32_t *d_status, void *stream) {
  if (!ctx || !d_packed || !d_out_rows || !d_status || !d_row_index) return HIMG_ERR_ARG;
  const himg_hip_ctx::HeadToken t = ctx->head;
  if (!t.valid)
    return fail(ctx, HIMG_ERR_ARG, "decode_rows_after_head_device: no decode_head_device in front of it (or another decode since)");
  if (t.w != width || t.h != height || t.c != num_channels || t.packed != d_packed || t.size != packed_size ||
      t.stream != stream)
    return fail(ctx, HIMG_ERR_ARG, "decode_rows_after_head_device: not the stream / geometry / HIP stream of decode_head_device");
  const int rc = decode_rows_indexed(ctx, d_packed, packed_size, width, height, num_channels, row0, row1, d_row_index,
                                     d_out_rows, d_status, stream, himg_dev::kDecRows);
  if (rc == HIMG_OK) ctx->head = t;   // (another row range of the same frame may follow)
  return rc;
}

// The row index of a stream in HBM by the header walk ALONE (k_dec_rowwalk finds the FRES
// payload itself), on the context's side stream behind whatever `stream` holds at the time of
// the call: launched in front of decode_head_device it runs BESIDE the head phase (they
// touch disjoint fields, as in every decode).  Results arrive with
// himg_hip_decode_walk_wait: d_row_index ([rows] offsets, [rows] lengths), d_rows_first,
// d_status (the walk's verdict only; the container's is the head phase's).
extern "C" int himg_hip_decode_walk_device(himg_hip_ctx *ctx, const void *d_packed, uint32_t packed_size,
                                           int width, int height, int num_channels, uint32_t *d_row_index,
                                           uint32_t *d_rows_first, int32_t *d_status, void *stream) {
  if (!ctx || !d_packed || !d_row_index || !d_rows_first || !d_status) return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, num_channels, num_channels, 1, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  g.fix_t2 = ctx->fix_t2;
  if (g.rows + 1 > 65535) return fail(ctx, HIMG_ERR_UNSUPPORTED, "grid too large");
  if ((uintptr_t)d_packed & 15) return fail(ctx, HIMG_ERR_ARG, "buffers must be 16-byte aligned");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dec_ws(ctx, g, 1);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  ctx->last_stream = s;
  rc = stage_sizes(ctx, &packed_size, 1, s);
  if (rc) return rc;
  hipStream_t w = ctx->use_side ? ctx->dstr.side : s;
  if (w != s) {
    HIP_TRY(ctx, hipEventRecord(ctx->dstr.ev_fork, s));
    HIP_TRY(ctx, hipStreamWaitEvent(w, ctx->dstr.ev_fork, 0));
  }
  himg_dev::launch_rowwalk_only(g, ctx->dec_ws, (const uint8_t *)d_packed, ((size_t)packed_size + 3) / 4 * 4,
                                (const uint32_t *)ctx->d_sizes.p, w);
  HIP_TRY(ctx, hipMemcpyAsync(d_row_index, ctx->dec_ws.row_off, (size_t)g.rows * 4, hipMemcpyDeviceToDevice, w));
  HIP_TRY(ctx, hipMemcpyAsync(d_row_index + g.rows, ctx->dec_ws.row_len, (size_t)g.rows * 4, hipMemcpyDeviceToDevice, w));
  HIP_TRY(ctx, hipMemcpyAsync(d_rows_first, &ctx->dec_ws.frames[0].rows_first, 4, hipMemcpyDeviceToDevice, w));
  HIP_TRY(ctx, hipMemcpyAsync(d_status, &ctx->dec_ws.frames[0].walk_status, 4, hipMemcpyDeviceToDevice, w));
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_decode_walk_ranges_device(himg_hip_ctx *ctx, const void *d_packed, uint32_t packed_size,
                                                  int width, int height, int num_channels, const int *range_end,
                                                  int n_ranges, uint32_t *d_row_index, uint32_t *d_rows_first,
                                                  int32_t *d_range_status, void *stream) {
  if (!ctx || !d_packed || !d_row_index || !d_rows_first || !d_range_status || !range_end || n_ranges < 1 ||
      n_ranges > HIMG_MAX_WALK_RANGES)
    return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, num_channels, num_channels, 1, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  g.fix_t2 = ctx->fix_t2;
  if (g.rows + 1 > 65535) return fail(ctx, HIMG_ERR_UNSUPPORTED, "grid too large");
  if ((uintptr_t)d_packed & 15) return fail(ctx, HIMG_ERR_ARG, "buffers must be 16-byte aligned");
  for (int k = 0; k < n_ranges; ++k)
    if (range_end[k] < 0 || (k && range_end[k] < range_end[k - 1])) return fail(ctx, HIMG_ERR_ARG, "range ends must ascend");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dec_ws(ctx, g, 1);
  if (rc) return rc;
  for (int k = 0; k < n_ranges; ++k)
    if (!ctx->ev_range[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_range[k], hipEventDisableTiming));
  hipStream_t s = (hipStream_t)stream;
  ctx->last_stream = s;
  rc = stage_sizes(ctx, &packed_size, 1, s);
  if (rc) return rc;
  hipStream_t w = ctx->use_side ? ctx->dstr.side : s;
  if (w != s) {
    HIP_TRY(ctx, hipEventRecord(ctx->dstr.ev_fork, s));
    HIP_TRY(ctx, hipStreamWaitEvent(w, ctx->dstr.ev_fork, 0));
  }
  int done = 0;   // rows whose index entries have been copied out
  for (int k = 0; k < n_ranges; ++k) {
    const bool last = k + 1 == n_ranges || range_end[k] >= g.rows;
    const int upto = last ? g.rows : range_end[k];
    himg_dev::launch_rowwalk_range(g, ctx->dec_ws, (const uint8_t *)d_packed, ((size_t)packed_size + 3) / 4 * 4,
                                   (const uint32_t *)ctx->d_sizes.p, last ? 0x7fffffff : upto, k > 0, w);
    if (upto > done) {
      HIP_TRY(ctx, hipMemcpyAsync(d_row_index + done, ctx->dec_ws.row_off + done, (size_t)(upto - done) * 4, hipMemcpyDeviceToDevice, w));
      HIP_TRY(ctx, hipMemcpyAsync(d_row_index + g.rows + done, ctx->dec_ws.row_len + done, (size_t)(upto - done) * 4, hipMemcpyDeviceToDevice, w));
      done = upto;
    }
    if (k == 0) HIP_TRY(ctx, hipMemcpyAsync(d_rows_first, &ctx->dec_ws.frames[0].rows_first, 4, hipMemcpyDeviceToDevice, w));
    HIP_TRY(ctx, hipMemcpyAsync(d_range_status + k, &ctx->dec_ws.frames[0].walk_status, 4, hipMemcpyDeviceToDevice, w));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_range[k], w));
    ctx->n_ranges = k + 1;
    if (last) {   // (ranges behind the last row: nothing left to walk; their events are this one)
      for (int j = k + 1; j < n_ranges; ++j) {
        HIP_TRY(ctx, hipMemcpyAsync(d_range_status + j, &ctx->dec_ws.frames[0].walk_status, 4, hipMemcpyDeviceToDevice, w));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_range[j], w));
      }
      ctx->n_ranges = n_ranges;
      break;
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_decode_walk_wait_range(himg_hip_ctx *ctx, int k) {
  if (!ctx || k < 0 || k >= ctx->n_ranges || !ctx->ev_range[k]) return HIMG_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev_range[k]));
  return HIMG_OK;
}

extern "C" int himg_hip_decode_walk_wait(himg_hip_ctx *ctx) {
  if (!ctx) return HIMG_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // (without a side stream the walk went to the caller's stream)
  HIP_TRY(ctx, hipStreamSynchronize(ctx->use_side ? ctx->dstr.side : ctx->last_stream));
  return HIMG_OK;
}

// Where the first FRES row header lies (container parse + the length of the serialised
// tree, no header walk): what the rank that holds a stream in HBM needs to send the head
// of the stream on its way before it indexes the rows.
extern "C" int himg_hip_decode_first_device(himg_hip_ctx *ctx, const void *d_packed, uint32_t packed_size,
                                            int width, int height, int num_channels, uint32_t *d_rows_first,
                                            int32_t *d_status, void *stream) {
  if (!ctx || !d_packed || !d_rows_first || !d_status) return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, num_channels, num_channels, 1, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  g.fix_t2 = ctx->fix_t2;
  if (g.rows + 1 > 65535) return fail(ctx, HIMG_ERR_UNSUPPORTED, "grid too large");
  if ((uintptr_t)d_packed & 15) return fail(ctx, HIMG_ERR_ARG, "buffers must be 16-byte aligned");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dec_ws(ctx, g, 1);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  ctx->last_stream = s;
  rc = stage_sizes(ctx, &packed_size, 1, s);
  if (rc) return rc;
  launch_decode(g, ctx->dec_ws, 1, (const uint8_t *)d_packed, ((size_t)packed_size + 3) / 4 * 4,
                (const uint32_t *)ctx->d_sizes.p, nullptr, d_status, s, &ctx->prof, ctx->allow_fused,
                nullptr, 0, 0, nullptr, true);
  HIP_TRY(ctx, hipMemcpyAsync(d_rows_first, &ctx->dec_ws.frames[0].rows_first, 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

// The same index on the host, for a stream in host memory (no GPU): chunk search
// (decoder.cpp:428-461), length of the serialised FRES tree (huffman_dec.cpp:152-229),
// row headers (huffman_dec.cpp:232-248).  A serial walk over a few thousand headers is
// microseconds on a CPU and 1 ms of dependent loads on the GPU.
static bool host_find_chunk(const uint8_t *p, size_t n, size_t *idx, uint32_t tag, uint32_t *size) {
  for (;;) {
    if (*idx + 8 > n) return false;
    uint32_t t, sz;
    memcpy(&t, p + *idx, 4);
    memcpy(&sz, p + *idx + 4, 4);
    *idx += 8;
    if (sz > 0x7fffffffu || *idx + sz > n) return false;
    if (t == tag) { *size = sz; return true; }
    *idx += sz;
  }
}

extern "C" int himg_hip_index_host(const uint8_t *packed, size_t packed_size, int fix_t2, int *width,
                                   int *height, int *num_channels, uint32_t *row_index,
                                   size_t index_rows, uint32_t *rows_first) {
  if (!packed || !width || !height || !num_channels || !rows_first) return HIMG_ERR_ARG;
  int rc = himg_hip_peek(packed, packed_size, width, height, num_channels);
  if (rc) return rc;
  const size_t rows = ((size_t)*height + 7) / 8;
  if (!row_index || index_rows < rows) return HIMG_ERR_CAPACITY;
  static const uint32_t tags[6] = {0x544d5246u, 0x50414d4cu, 0x5345524cu, 0x47464351u, 0x50414d46u, 0x53455246u};
  size_t idx = 12;
  uint32_t sz = 0;
  for (int t = 0; t < 6; ++t) {
    if (!host_find_chunk(packed, packed_size, &idx, tags[t], &sz)) return HIMG_ERR_FORMAT;
    if (t < 5) idx += sz;
  }
  const size_t coff = idx, end = idx + sz;
  size_t bit = 0;
  {
    const size_t bit_end = 8 * (size_t)(sz < (uint32_t)kTreeStride ? sz : (uint32_t)kTreeStride);
    int open = 1, count = 0;
    while (open > 0) {
      if (count >= 2 * kNumSym - 1 || bit >= bit_end) return HIMG_ERR_FORMAT;
      ++count;
      if ((packed[coff + (bit >> 3)] >> (bit & 7)) & 1) {
        if (bit + 10 > bit_end) return HIMG_ERR_FORMAT;
        bit += 10;
        --open;
      } else {
        bit += 1;
        ++open;
      }
    }
  }
  size_t q = coff + ((bit + 7) >> 3);
  if (q >= end) return HIMG_ERR_FORMAT;
  *rows_first = (uint32_t)q;
  if (fix_t2 && rows == 1) {
    row_index[0] = (uint32_t)q;
    row_index[rows] = (uint32_t)(end - q);
    return HIMG_OK;
  }
  size_t r = 0;
  while (q != end) {
    if (q + 2 > end) return HIMG_ERR_FORMAT;
    uint32_t len = packed[q] | (packed[q + 1] << 8);
    q += 2;
    if (len & 0x8000u) {
      if (q + 2 > end) return HIMG_ERR_FORMAT;
      len = (len & 0x7fffu) | ((uint32_t)(packed[q] | (packed[q + 1] << 8)) << 15);
      q += 2;
    }
    if (len > end - q) return HIMG_ERR_FORMAT;
    if (r < rows) { row_index[r] = (uint32_t)q; row_index[rows + r] = len; }
    ++r;
    q += len;
  }
  return r < rows ? HIMG_ERR_FORMAT : HIMG_OK;
}

// ---------------------------------------------------------------------------
// Host-buffer API.
// ---------------------------------------------------------------------------
// ---- host-buffer API ---------------------------------------------------------
// The stream / the pixels of the last host call stay resident in ctx->h_out; the
// entry points differ only in where they copy them to.

static int encode_core(himg_hip_ctx *ctx, const uint8_t *data, int width, int height,
                       int pixel_stride, int num_channels, int quality, int use_ycbcr,
                       uint32_t *n_out) {
  Geom g;
  if (!make_geom(width, height, pixel_stride, num_channels, use_ycbcr, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t cap = himg_hip_max_packed_size(width, height, num_channels);
  if (!ctx->h_in.reserve(round_up((size_t)g.frame_bytes, 256)) || !ctx->h_out.reserve(cap) ||
      !ctx->h_sizes.reserve(256) || !ctx->h_status.reserve(256))
    return fail(ctx, HIMG_ERR_HIP, "staging allocation failed");
  ctx->host_bytes = 0;
  HIP_TRY(ctx, hipMemcpy(ctx->h_in.p, data, (size_t)g.frame_bytes, hipMemcpyHostToDevice));
  int rc = himg_hip_encode_device(ctx, ctx->h_in.p, 1, width, height, pixel_stride, num_channels,
                                  quality, use_ycbcr, ctx->h_out.p, cap, (uint32_t *)ctx->h_sizes.p,
                                  (int32_t *)ctx->h_status.p, nullptr);
  if (rc) return rc;
  uint32_t n = 0;
  int32_t st = 0;
  HIP_TRY(ctx, hipMemcpy(&n, ctx->h_sizes.p, 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(&st, ctx->h_status.p, 4, hipMemcpyDeviceToHost));
  if (st) return fail(ctx, status_to_code(st), "device encode reported an error");
  ctx->host_bytes = n;
  *n_out = n;
  return HIMG_OK;
}

extern "C" int himg_hip_encode(himg_hip_ctx *ctx, const uint8_t *data, int width, int height,
                               int pixel_stride, int num_channels, int quality, int use_ycbcr,
                               uint8_t **out, size_t *out_size) {
  if (!ctx || !data || !out || !out_size) return HIMG_ERR_ARG;
  *out = nullptr;
  *out_size = 0;
  uint32_t n = 0;
  const int rc = encode_core(ctx, data, width, height, pixel_stride, num_channels, quality, use_ycbcr, &n);
  if (rc) return rc;
  uint8_t *buf = (uint8_t *)std::malloc(n ? n : 1);
  if (!buf) return fail(ctx, HIMG_ERR_ARG, "out of host memory");
  HIP_TRY(ctx, hipMemcpy(buf, ctx->h_out.p, n, hipMemcpyDeviceToHost));
  *out = buf;
  *out_size = n;
  return HIMG_OK;
}

extern "C" int himg_hip_encode_to(himg_hip_ctx *ctx, const uint8_t *data, int width, int height,
                                  int pixel_stride, int num_channels, int quality, int use_ycbcr,
                                  uint8_t *dst, size_t dst_cap, size_t *out_size) {
  if (!ctx || !data || !out_size) return HIMG_ERR_ARG;
  *out_size = 0;
  uint32_t n = 0;
  const int rc = encode_core(ctx, data, width, height, pixel_stride, num_channels, quality, use_ycbcr, &n);
  if (rc) return rc;
  *out_size = n;
  if (!dst || dst_cap < n) return fail(ctx, HIMG_ERR_CAPACITY, "output buffer too small");
  HIP_TRY(ctx, hipMemcpy(dst, ctx->h_out.p, n, hipMemcpyDeviceToHost));
  return HIMG_OK;
}

extern "C" int himg_hip_fetch_last(himg_hip_ctx *ctx, uint8_t *dst, size_t dst_cap, size_t *size) {
  if (!ctx || !size) return HIMG_ERR_ARG;
  *size = ctx->host_bytes;
  if (!ctx->host_bytes) return fail(ctx, HIMG_ERR_ARG, "no result to fetch");
  if (!dst || dst_cap < ctx->host_bytes) return fail(ctx, HIMG_ERR_CAPACITY, "output buffer too small");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpy(dst, ctx->h_out.p, ctx->host_bytes, hipMemcpyDeviceToHost));
  return HIMG_OK;
}

// Geometry from the FRMT chunk (decoder.cpp:144-200).  Returns nullptr or the
// reference's message for the failing check.
static const char *parse_header(const uint8_t *packed, size_t packed_size, int *W, int *H, int *C) {
  // decoder.cpp:144-166: the RIFF size field must match the buffer exactly.
  if (packed_size < 12 || packed_size > 0x7fffffffu || memcmp(packed, "RIFF", 4) != 0 ||
      memcmp(packed + 8, "HIMG", 4) != 0 ||
      (size_t)(packed[4] | (packed[5] << 8) | (packed[6] << 16) | ((uint32_t)packed[7] << 24)) + 8 != packed_size)
    return "Not a RIFF HIMG file.\n";
  size_t idx = 12;
  for (;;) {
    if (idx + 8 > packed_size) return "Error decoding header.\n";
    const uint32_t sz = packed[idx + 4] | (packed[idx + 5] << 8) | (packed[idx + 6] << 16) |
                        ((uint32_t)packed[idx + 7] << 24);
    const bool frmt = memcmp(packed + idx, "FRMT", 4) == 0;
    idx += 8;
    if (idx + sz > packed_size) return "Error decoding header.\n";
    if (frmt) {
      if (sz < 11 || packed[idx] != 1) return "Error decoding header.\n";
      *W = (int)(packed[idx + 1] | (packed[idx + 2] << 8) | (packed[idx + 3] << 16) | ((uint32_t)packed[idx + 4] << 24));
      *H = (int)(packed[idx + 5] | (packed[idx + 6] << 8) | (packed[idx + 7] << 16) | ((uint32_t)packed[idx + 8] << 24));
      *C = packed[idx + 9];
      return nullptr;
    }
    idx += sz;
  }
}

extern "C" int himg_hip_peek(const uint8_t *packed, size_t packed_size, int *width, int *height,
                             int *num_channels) {
  if (!packed || !width || !height || !num_channels) return HIMG_ERR_ARG;
  int W = 0, H = 0, C = 0;
  if (parse_header(packed, packed_size, &W, &H, &C)) return HIMG_ERR_FORMAT;
  // The FRMT fields are untrusted: callers size their output from them, so apply
  // the engine's limits (make_geom) before anybody allocates.
  Geom g;
  if (!make_geom(W, H, C, C, 1, &g)) return HIMG_ERR_UNSUPPORTED;
  *width = W; *height = H; *num_channels = C;
  return HIMG_OK;
}

static int decode_core(himg_hip_ctx *ctx, const uint8_t *packed, size_t packed_size, int *W, int *H,
                       int *C) {
  // The host only needs the geometry to size the launch; every check is repeated
  // on the device (k_dec_parse).
  if (const char *msg = parse_header(packed, packed_size, W, H, C)) return fail(ctx, HIMG_ERR_FORMAT, msg);
  Geom g;
  if (!make_geom(*W, *H, *C, *C, 1, &g)) return fail(ctx, HIMG_ERR_UNSUPPORTED, "unsupported geometry");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t in_cap = round_up(packed_size + 16, 256);
  const size_t out_bytes = (size_t)*W * *H * *C;
  if (!ctx->h_in.reserve(in_cap) || !ctx->h_out.reserve(round_up(out_bytes, 256)) ||
      !ctx->h_status.reserve(256))
    return fail(ctx, HIMG_ERR_HIP, "staging allocation failed");
  ctx->host_bytes = 0;
  // (The row kernels read whole dwords and a few dwords ahead: nothing of the stream decoded
  // before may lie behind this one.)
  // (Everything below is ordered on the null stream, which the decode runs on; the one wait of
  // this call is the read of the verdict at the end.)
  HIP_TRY(ctx, hipMemsetAsync((uint8_t *)ctx->h_in.p + (packed_size & ~(size_t)15), 0, in_cap - (packed_size & ~(size_t)15), nullptr));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_in.p, packed, packed_size, hipMemcpyHostToDevice, nullptr));
  const uint32_t sz32 = (uint32_t)packed_size;
  // The stream is in host memory: the FRES rows are indexed HERE -- the walk over the row
  // size headers (huffman_dec.cpp:232-248) is a chain of dependent reads, microseconds on
  // a CPU and 0.24 ms of a 0.41 ms decode as k_dec_rowwalk's 512 dependent HBM loads -- and
  // the index goes up with the stream.  A stream the host walk does not accept (damaged
  // headers, a geometry it does not index) takes the device walk, which words the verdict.
  int rc = -1;
  {
    // The index is written into a pinned buffer the context keeps (this call returns only after
    // the decode that reads its copy has finished) and goes up asynchronously.
    const size_t n_idx = 2 * (size_t)g.rows;
    if (ctx->hp_index_cap < n_idx) {
      if (ctx->hp_index) hipHostFree(ctx->hp_index);
      ctx->hp_index = nullptr;
      ctx->hp_index_cap = 0;
      // (An error return from here on first waits for the uploads above: the caller may free or
      // overwrite `packed` the moment this call returns.)
      if (hipHostMalloc((void **)&ctx->hp_index, round_up(n_idx * 4, 4096), hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamSynchronize(nullptr);
        return fail(ctx, HIMG_ERR_HIP, "pinned index allocation failed");
      }
      ctx->hp_index_cap = round_up(n_idx * 4, 4096) / 4;
    }
    uint32_t first = 0;
    int w2 = 0, h2 = 0, c2 = 0;
    if (g.rows >= 2 && ctx->h_index.reserve(round_up(n_idx * 4, 256)) &&
        himg_hip_index_host(packed, packed_size, ctx->fix_t2, &w2, &h2, &c2, ctx->hp_index, (size_t)g.rows, &first) == HIMG_OK) {
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_index.p, ctx->hp_index, n_idx * 4, hipMemcpyHostToDevice, nullptr));
      rc = himg_hip_decode_rows_indexed_device(ctx, ctx->h_in.p, sz32, *W, *H, *C, 0, g.rows, (const uint32_t *)ctx->h_index.p,
                                               ctx->h_out.p, (int32_t *)ctx->h_status.p, nullptr);
    }
  }
  if (rc == -1)
    rc = himg_hip_decode_device(ctx, ctx->h_in.p, in_cap, &sz32, 1, *W, *H, *C, ctx->h_out.p,
                                (int32_t *)ctx->h_status.p, nullptr);
  if (rc) {
    (void)hipStreamSynchronize(nullptr);   // the stream / index uploads may still be reading the caller's and the pinned buffer
    return rc;
  }
  int32_t st = 0;
  HIP_TRY(ctx, hipMemcpy(&st, ctx->h_status.p, 4, hipMemcpyDeviceToHost));
  if (st) {
    const int code = status_to_code(st);
    if (code == HIMG_ERR_FORMAT) {
      ctx->err = format_message(st);
      return code;
    }
    return fail(ctx, code, "device decode reported an error");
  }
  ctx->host_bytes = out_bytes;
  return HIMG_OK;
}

extern "C" int himg_hip_decode(himg_hip_ctx *ctx, const uint8_t *packed, size_t packed_size,
                               uint8_t **out, int *width, int *height, int *num_channels) {
  if (!ctx || !packed || !out || !width || !height || !num_channels) return HIMG_ERR_ARG;
  *out = nullptr;
  int W = 0, H = 0, C = 0;
  const int rc = decode_core(ctx, packed, packed_size, &W, &H, &C);
  if (rc) return rc;
  const size_t out_bytes = ctx->host_bytes;
  uint8_t *buf = (uint8_t *)std::malloc(out_bytes ? out_bytes : 1);
  if (!buf) return fail(ctx, HIMG_ERR_ARG, "out of host memory");
  HIP_TRY(ctx, hipMemcpy(buf, ctx->h_out.p, out_bytes, hipMemcpyDeviceToHost));
  *out = buf;
  *width = W; *height = H; *num_channels = C;
  return HIMG_OK;
}

extern "C" int himg_hip_decode_to(himg_hip_ctx *ctx, const uint8_t *packed, size_t packed_size,
                                  uint8_t *dst, size_t dst_cap, int *width, int *height,
                                  int *num_channels) {
  if (!ctx || !packed || !width || !height || !num_channels) return HIMG_ERR_ARG;
  int W = 0, H = 0, C = 0;
  const int rc = decode_core(ctx, packed, packed_size, &W, &H, &C);
  if (rc) return rc;
  *width = W; *height = H; *num_channels = C;
  if (!dst || dst_cap < ctx->host_bytes) return fail(ctx, HIMG_ERR_CAPACITY, "output buffer too small");
  HIP_TRY(ctx, hipMemcpy(dst, ctx->h_out.p, ctx->host_bytes, hipMemcpyDeviceToHost));
  return HIMG_OK;
}

// ---- batched host API: frames in flight ------------------------------------------

static int pipe_init(himg_hip_ctx *ctx) {
  himg_hip_ctx::Pipe &p = ctx->pipe;
  if (p.ready) return HIMG_OK;
  HIP_TRY(ctx, hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking));
  HIP_TRY(ctx, hipStreamCreateWithFlags(&p.s_comp, hipStreamNonBlocking));
  HIP_TRY(ctx, hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking));
  for (int k = 0; k < 2; ++k) {
    HIP_TRY(ctx, hipEventCreateWithFlags(&p.ev_in[k], hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&p.ev_k[k], hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&p.ev_out[k], hipEventDisableTiming));
    if (!p.meta[k].reserve(256)) return fail(ctx, HIMG_ERR_HIP, "staging allocation failed");
  }
  HIP_TRY(ctx, hipHostMalloc((void **)&p.h_meta, 64, hipHostMallocDefault));
  p.ready = true;
  return HIMG_OK;
}

extern "C" int himg_hip_encode_batch(himg_hip_ctx *ctx, const uint8_t *const *frames, int n, int width,
                                     int height, int pixel_stride, int num_channels, int quality,
                                     int use_ycbcr, uint8_t *const *dst, const size_t *dst_cap,
                                     size_t *out_sizes) {
  if (!ctx || !frames || !dst || !dst_cap || !out_sizes || n < 0) return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, pixel_stride, num_channels, use_ycbcr, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = pipe_init(ctx);
  if (rc) return rc;
  himg_hip_ctx::Pipe &p = ctx->pipe;
  const size_t cap = himg_hip_max_packed_size(width, height, num_channels);
  for (int k = 0; k < 2; ++k)
    if (!p.in[k].reserve(round_up((size_t)g.frame_bytes, 256)) || !p.out[k].reserve(cap))
      return fail(ctx, HIMG_ERR_HIP, "staging allocation failed");
  ctx->host_bytes = 0;
  int first_err = HIMG_OK;
  // Fetch frame j's verdict and start the copy of its stream.
  auto finish = [&](int j) -> int {
    const int slot = j & 1;
    out_sizes[j] = 0;
    HIP_TRY(ctx, hipEventSynchronize(p.ev_k[slot]));
    const uint32_t nbytes = p.h_meta[slot * 4 + 0];
    const int32_t st = (int32_t)p.h_meta[slot * 4 + 1];
    int err = HIMG_OK;
    if (st) err = fail(ctx, status_to_code(st), "device encode reported an error");
    else if (!dst[j] || dst_cap[j] < nbytes) err = fail(ctx, HIMG_ERR_CAPACITY, "output buffer too small");
    if (!err) {
      HIP_TRY(ctx, hipMemcpyAsync(dst[j], p.out[slot].p, nbytes, hipMemcpyDeviceToHost, p.s_out));
      out_sizes[j] = nbytes;
    } else if (!first_err) {
      first_err = err;
    }
    HIP_TRY(ctx, hipEventRecord(p.ev_out[slot], p.s_out));
    return HIMG_OK;
  };
  for (int i = 0; i < n; ++i) {
    const int slot = i & 1;
    if (!frames[i]) return fail(ctx, HIMG_ERR_ARG, "null frame");
    if (i >= 2) HIP_TRY(ctx, hipEventSynchronize(p.ev_out[slot]));   // staging of frame i-2 is free again
    HIP_TRY(ctx, hipMemcpyAsync(p.in[slot].p, frames[i], (size_t)g.frame_bytes, hipMemcpyHostToDevice, p.s_in));
    HIP_TRY(ctx, hipEventRecord(p.ev_in[slot], p.s_in));
    HIP_TRY(ctx, hipStreamWaitEvent(p.s_comp, p.ev_in[slot], 0));
    rc = himg_hip_encode_device(ctx, p.in[slot].p, 1, width, height, pixel_stride, num_channels, quality,
                                use_ycbcr, p.out[slot].p, cap, (uint32_t *)p.meta[slot].p,
                                (int32_t *)p.meta[slot].p + 1, p.s_comp);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(p.h_meta + slot * 4, p.meta[slot].p, 8, hipMemcpyDeviceToHost, p.s_comp));
    HIP_TRY(ctx, hipEventRecord(p.ev_k[slot], p.s_comp));
    if (i >= 1 && (rc = finish(i - 1))) return rc;
  }
  if (n >= 1 && (rc = finish(n - 1))) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(p.s_out));
  return first_err;
}

extern "C" int himg_hip_decode_batch(himg_hip_ctx *ctx, const uint8_t *const *packed,
                                     const size_t *packed_sizes, int n, uint8_t *const *dst,
                                     const size_t *dst_cap, int *widths, int *heights, int *channels) {
  if (!ctx || !packed || !packed_sizes || !dst || !dst_cap || !widths || !heights || !channels || n < 0)
    return HIMG_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = pipe_init(ctx);
  if (rc) return rc;
  himg_hip_ctx::Pipe &p = ctx->pipe;
  ctx->host_bytes = 0;
  int first_err = HIMG_OK;
  std::vector<size_t> out_bytes((size_t)n, 0);
  std::vector<int> launched((size_t)n, 0);
  auto finish = [&](int j) -> int {
    const int slot = j & 1;
    if (launched[j]) {
      HIP_TRY(ctx, hipEventSynchronize(p.ev_k[slot]));
      const int32_t st = (int32_t)p.h_meta[slot * 4 + 1];
      int err = HIMG_OK;
      if (st) {
        err = status_to_code(st);
        if (err == HIMG_ERR_FORMAT) ctx->err = format_message(st);
        else fail(ctx, err, "device decode reported an error");
      } else if (!dst[j] || dst_cap[j] < out_bytes[j]) {
        err = fail(ctx, HIMG_ERR_CAPACITY, "output buffer too small");
      }
      if (!err) HIP_TRY(ctx, hipMemcpyAsync(dst[j], p.out[slot].p, out_bytes[j], hipMemcpyDeviceToHost, p.s_out));
      else { widths[j] = heights[j] = channels[j] = 0; if (!first_err) first_err = err; }
    }
    HIP_TRY(ctx, hipEventRecord(p.ev_out[slot], p.s_out));
    return HIMG_OK;
  };
  for (int i = 0; i < n; ++i) {
    const int slot = i & 1;
    widths[i] = heights[i] = channels[i] = 0;
    if (i >= 2) HIP_TRY(ctx, hipEventSynchronize(p.ev_out[slot]));
    int W = 0, H = 0, C = 0;
    Geom g;
    const char *msg = packed[i] ? parse_header(packed[i], packed_sizes[i], &W, &H, &C) : "Not a RIFF HIMG file.\n";
    if (msg || !make_geom(W, H, C, C, 1, &g)) {
      if (!first_err) first_err = msg ? fail(ctx, HIMG_ERR_FORMAT, msg) : fail(ctx, HIMG_ERR_UNSUPPORTED, "unsupported geometry");
    } else {
      const size_t in_cap = round_up(packed_sizes[i] + 16, 256);
      out_bytes[i] = (size_t)W * H * C;
      if (!p.in[slot].reserve(in_cap) || !p.out[slot].reserve(round_up(out_bytes[i], 256)))
        return fail(ctx, HIMG_ERR_HIP, "staging allocation failed");
      HIP_TRY(ctx, hipMemcpyAsync(p.in[slot].p, packed[i], packed_sizes[i], hipMemcpyHostToDevice, p.s_in));
      HIP_TRY(ctx, hipEventRecord(p.ev_in[slot], p.s_in));
      HIP_TRY(ctx, hipStreamWaitEvent(p.s_comp, p.ev_in[slot], 0));
      p.h_meta[8 + slot] = (uint32_t)packed_sizes[i];   // pinned: stays valid until the copy has run
      rc = himg_hip_decode_device(ctx, p.in[slot].p, in_cap, p.h_meta + 8 + slot, 1, W, H, C, p.out[slot].p,
                                  (int32_t *)p.meta[slot].p + 1, p.s_comp);
      if (rc) return rc;
      HIP_TRY(ctx, hipMemcpyAsync(p.h_meta + slot * 4, p.meta[slot].p, 8, hipMemcpyDeviceToHost, p.s_comp));
      HIP_TRY(ctx, hipEventRecord(p.ev_k[slot], p.s_comp));
      widths[i] = W; heights[i] = H; channels[i] = C;
      launched[i] = 1;
    }
    if (i >= 1 && (rc = finish(i - 1))) return rc;
  }
  if (n >= 1 && (rc = finish(n - 1))) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(p.s_out));
  return first_err;
}

// ---------------------------------------------------------------------------
// Row-sharded encode of one frame over several GPUs (one context per rank).
// ---------------------------------------------------------------------------
extern "C" int himg_hip_shard_stats(himg_hip_ctx *ctx, const void *d_frame_base, int width,
                                    int height, int pixel_stride, int num_channels, int quality,
                                    int use_ycbcr, int row0, int row1, uint32_t *d_fres_hist,
                                    uint8_t *d_low_rows, void *stream) {
  if (!ctx || !d_frame_base || !d_fres_hist || !d_low_rows) return HIMG_ERR_ARG;
  Geom g;
  if (!make_geom(width, height, pixel_stride, num_channels, use_ycbcr, &g))
    return fail(ctx, HIMG_ERR_ARG, "bad geometry");
  // row0 == row1: a rank without rows (more ranks than low-res macro rows) still
  // takes part in every phase; it contributes an all-zero histogram.
  if (row0 < 0 || row1 > g.rows || row0 > row1 || g.rows > 65535)
    return fail(ctx, HIMG_ERR_ARG, "bad block-row range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // A range of a thousand block rows and more goes through the token stream (k_tok / k_emit_tok: a
  // wavefront per row packs it in ~0.3 ms whatever the number of rows; below, the 1024-lane kernels
  // over the dense plane are faster); HIMG_OPT_ROW_TOKENS forces either.
  const bool shard_tok = ctx->row_tokens != 0 && (ctx->row_tokens > 0 || row1 - row0 >= 1024);
  int rc = ensure_enc_ws(ctx, g, 1, shard_tok, shard_tok);
  if (rc) return rc;
  auto &sh = ctx->shard;
  rc = build_static(g, quality, &sh.sc, &sh.st, &sh.lt);
  if (rc) return fail(ctx, rc, "unsupported table configuration");
  sh.g = g; sh.r0 = row0; sh.r1 = row1; sh.valid = true;
  hipStream_t s = (hipStream_t)stream;
  if (row1 > row0) {
    launch_shard_stats(g, ctx->enc_ws, (const uint8_t *)d_frame_base, sh.st,
                       (const uint8_t *)ctx->fmap_lut.p, row0, row1, s, &ctx->prof);
  } else {
    HIP_TRY(ctx, hipMemsetAsync(ctx->enc_ws.hist, 0, 2 * kHistStride * sizeof(uint32_t), s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->enc_ws.status, 0, sizeof(int32_t), s));
  }
  HIP_TRY(ctx, hipMemcpyAsync(d_fres_hist, ctx->enc_ws.hist + kHistStride, kNumSym * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, s));
  const size_t n = (size_t)(row1 - row0) * g.cols;
  for (int c = 0; c < g.C; ++c)
    HIP_TRY(ctx, hipMemcpyAsync(d_low_rows + (size_t)c * n,
                                ctx->enc_ws.low + ((size_t)c * g.rows + row0) * g.cols, n,
                                hipMemcpyDeviceToDevice, s));
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_shard_row_bits(himg_hip_ctx *ctx, const uint32_t *d_fres_hist_global,
                                       uint32_t *d_row_bits, void *stream) {
  if (!ctx || !ctx->shard.valid || !d_fres_hist_global || !d_row_bits) return HIMG_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  auto &sh = ctx->shard;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->enc_ws.hist + kHistStride, d_fres_hist_global,
                              kNumSym * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  launch_shard_row_bits(sh.g, ctx->enc_ws, sh.r0, sh.r1, d_row_bits, s, &ctx->prof);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_shard_emit(himg_hip_ctx *ctx, const uint32_t *d_all_row_bits, void *d_rel,
                                   size_t rel_cap, uint32_t *d_rel_size, void *stream) {
  if (!ctx || !ctx->shard.valid || !d_all_row_bits || !d_rel || !d_rel_size) return HIMG_ERR_ARG;
  if ((rel_cap & 3) || ((uintptr_t)d_rel & 15)) return fail(ctx, HIMG_ERR_ARG, "bad relative buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  auto &sh = ctx->shard;
  hipStream_t s = (hipStream_t)stream;
  launch_shard_emit(sh.g, ctx->enc_ws, sh.sc, d_all_row_bits, (uint8_t *)d_rel, rel_cap, d_rel_size,
                    sh.r0, sh.r1, s, &ctx->prof);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_shard_assemble(himg_hip_ctx *ctx, const uint8_t *d_low_full,
                                       const uint32_t *d_all_row_bits, const void *d_rel,
                                       size_t rel_bytes, void *d_out, size_t out_cap,
                                       uint32_t *d_size, int32_t *d_status, void *stream) {
  if (!ctx || !ctx->shard.valid || !d_low_full || !d_all_row_bits || !d_rel || !d_out || !d_size)
    return HIMG_ERR_ARG;
  if ((out_cap & 255) || ((uintptr_t)d_out & 15)) return fail(ctx, HIMG_ERR_ARG, "bad output buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  auto &sh = ctx->shard;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->enc_ws.low, d_low_full, (size_t)sh.g.C * sh.g.rows * sh.g.cols,
                              hipMemcpyDeviceToDevice, s));
  launch_shard_assemble(sh.g, ctx->enc_ws, sh.sc, sh.lt, d_all_row_bits, (const uint8_t *)d_rel,
                        rel_bytes, (uint8_t *)d_out, out_cap, d_size, s, &ctx->prof);
  if (d_status)
    hipLaunchKernelGGL(k_copy_status, dim3(1), dim3(64), 0, s, ctx->enc_ws.status, d_status, 1);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_shard_head(himg_hip_ctx *ctx, const uint8_t *d_low_full, const uint32_t *d_all_row_bits,
                                   void *d_out, size_t out_cap, uint32_t *d_size, uint32_t *d_head,
                                   int32_t *d_status, void *stream) {
  if (!ctx || !ctx->shard.valid || !d_low_full || !d_all_row_bits || !d_out || !d_size || !d_head)
    return HIMG_ERR_ARG;
  if ((out_cap & 255) || ((uintptr_t)d_out & 15)) return fail(ctx, HIMG_ERR_ARG, "bad output buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  auto &sh = ctx->shard;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->enc_ws.low, d_low_full, (size_t)sh.g.C * sh.g.rows * sh.g.cols,
                              hipMemcpyDeviceToDevice, s));
  launch_shard_head(sh.g, ctx->enc_ws, sh.sc, sh.lt, d_all_row_bits, (uint8_t *)d_out, out_cap, d_size, d_head,
                    sh.r0, sh.r1, s, &ctx->prof);
  if (d_status)
    hipLaunchKernelGGL(k_copy_status, dim3(1), dim3(64), 0, s, ctx->enc_ws.status, d_status, 1);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

extern "C" int himg_hip_shard_finish(himg_hip_ctx *ctx, void *d_out, size_t out_cap, const uint32_t *d_size,
                                     void *stream) {
  if (!ctx || !ctx->shard.valid || !d_out || !d_size) return HIMG_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  launch_shard_finish(ctx->shard.g, ctx->enc_ws, (uint8_t *)d_out, out_cap, d_size, (hipStream_t)stream, &ctx->prof);
  HIP_TRY(ctx, hipGetLastError());
  return HIMG_OK;
}

// ---------------------------------------------------------------------------
// Introspection.
// ---------------------------------------------------------------------------
extern "C" int himg_hip_debug_read(himg_hip_ctx *ctx, int what, int frame, void *dst,
                                   size_t dst_bytes, size_t *written) {
  if (!ctx || !dst) return HIMG_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  const void *src = nullptr;
  size_t n = 0;
  const bool dec = (what & 0x100) != 0;  // 0x100 | what: decoder-side buffers
  what &= 0xff;
  if (what == HIMG_DBG_LOOP_COUNTS) {
    unsigned long long c[2 * himg_dev::kLoopCounters];
    if (dst_bytes < sizeof(c)) return HIMG_ERR_CAPACITY;
    if ((dec ? himg_dev::loop_counts_read_dec(c) : himg_dev::loop_counts_read_enc(c)) != 0) return HIMG_ERR_ARG;
    memcpy(dst, c, sizeof(c));
    if (written) *written = sizeof(c);
    return HIMG_OK;
  }
  if (!dec) {
    if (!ctx->enc_valid || frame < 0 || frame >= ctx->enc_batch) return HIMG_ERR_ARG;
    const Geom &g = ctx->enc_geom;
    const EncWs &w = ctx->enc_ws;
    const int nsp = g.lres_spans + g.rows;
    switch (what) {
      case HIMG_DBG_AVG: src = w.avg + frame * w.plane_stride; n = (size_t)g.C * g.rows * g.cols; break;
      case HIMG_DBG_LOWRES: src = w.low + frame * w.plane_stride; n = (size_t)g.C * g.rows * g.cols; break;
      case HIMG_DBG_LRES_SYM: src = w.lres_sym + frame * w.lres_stride; n = (size_t)g.lres_size; break;
      case HIMG_DBG_FRES_SYM: src = w.fres_sym + frame * w.fres_stride; n = (size_t)g.fres_size; break;
      case HIMG_DBG_FRES_TOK_SYM: {
        // The token stream of the last batch encode expanded into symbols again (a scratch plane).
        if (!w.tok) return HIMG_ERR_ARG;
        if (!ctx->e_tokx.reserve(round_up((size_t)g.fres_size + 16, 256))) return fail(ctx, HIMG_ERR_HIP, "scratch plane allocation failed");
        himg_dev::launch_tok_expand(g, w, frame, (uint8_t *)ctx->e_tokx.p, nullptr);
        HIP_TRY(ctx, hipDeviceSynchronize());
        src = ctx->e_tokx.p; n = (size_t)g.fres_size; break;
      }
      case HIMG_DBG_LRES_HIST: src = w.hist + ((size_t)frame * 2 + 0) * kHistStride; n = kNumSym * 4; break;
      case HIMG_DBG_FRES_HIST: src = w.hist + ((size_t)frame * 2 + 1) * kHistStride; n = kNumSym * 4; break;
      case HIMG_DBG_LRES_LEN: src = w.lens + ((size_t)frame * 2 + 0) * kHistStride; n = kNumSym * 4; break;
      case HIMG_DBG_FRES_LEN: src = w.lens + ((size_t)frame * 2 + 1) * kHistStride; n = kNumSym * 4; break;
      case HIMG_DBG_LRES_CODE: src = w.codes + ((size_t)frame * 2 + 0) * kHistStride; n = kNumSym * 8; break;
      case HIMG_DBG_FRES_CODE: src = w.codes + ((size_t)frame * 2 + 1) * kHistStride; n = kNumSym * 8; break;
      case HIMG_DBG_FRES_ROW_BYTES: {
        // Convert payload bits to bytes on the host.
        std::vector<uint32_t> bits(g.rows);
        HIP_TRY(ctx, hipMemcpy(bits.data(), w.span_bits + (size_t)frame * nsp + g.lres_spans,
                               (size_t)g.rows * 4, hipMemcpyDeviceToHost));
        n = (size_t)g.rows * 4;
        if (dst_bytes < n) return HIMG_ERR_CAPACITY;
        for (int r = 0; r < g.rows; ++r) ((uint32_t *)dst)[r] = (bits[r] + 7) >> 3;
        if (written) *written = n;
        return HIMG_OK;
      }
      default: return HIMG_ERR_ARG;
    }
  } else {
    if (!ctx->dec_valid || frame < 0 || frame >= ctx->dec_batch) return HIMG_ERR_ARG;
    const Geom &g = ctx->dec_geom;
    const DecWs &w = ctx->dec_ws;
    switch (what) {
      case HIMG_DBG_LOWRES: src = w.low + frame * w.plane_stride; n = (size_t)g.C * g.rows * g.cols; break;
      case HIMG_DBG_LRES_SYM: src = w.lres_sym + frame * w.lres_stride; n = (size_t)g.lres_size; break;
      case HIMG_DBG_FRES_SYM: src = w.fres_sym + frame * w.fres_stride; n = (size_t)g.fres_size; break;
      case HIMG_DBG_DEC_STATS: src = w.stats + (size_t)frame * (g.rows + 1) * 8; n = (size_t)(g.rows + 1) * 32; break;
      case HIMG_DBG_PARSE_STATS: src = w.parse_stats + (size_t)frame * 4; n = 16; break;
      case HIMG_DBG_ROWCOUNT_STATS: src = w.rc_stats + (size_t)frame * g.rows * 8; n = (size_t)g.rows * 32; break;
      default: return HIMG_ERR_ARG;
    }
  }
  if (dst_bytes < n) return HIMG_ERR_CAPACITY;
  HIP_TRY(ctx, hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
  if (written) *written = n;
  return HIMG_OK;
}

extern "C" int himg_hip_profile_enable(himg_hip_ctx *ctx, int enable) {
  if (!ctx) return HIMG_ERR_ARG;
  ctx->prof.enabled = enable != 0;
  return HIMG_OK;
}

extern "C" int himg_hip_profile_reset(himg_hip_ctx *ctx) {
  if (!ctx) return HIMG_ERR_ARG;
  hipSetDevice(ctx->device);
  ctx->prof.collect();
  ctx->prof.acc.clear();
  return HIMG_OK;
}

extern "C" int himg_hip_profile_read(himg_hip_ctx *ctx, int *n_stages,
                                     const char *names[HIMG_MAX_STAGES],
                                     double ms[HIMG_MAX_STAGES], int launches[HIMG_MAX_STAGES]) {
  if (!ctx || !n_stages) return HIMG_ERR_ARG;
  hipSetDevice(ctx->device);
  ctx->prof.collect();
  int n = 0;
  for (auto &a : ctx->prof.acc) {
    if (n >= HIMG_MAX_STAGES) break;
    names[n] = a.name.c_str();
    ms[n] = a.ms;
    launches[n] = a.n;
    ++n;
  }
  *n_stages = n;
  return HIMG_OK;
}
