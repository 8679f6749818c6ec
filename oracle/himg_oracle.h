/*
 * himg_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar, single-threaded restatement of
 * the reference HIMG codec (mbitsnbites/himg, src/lib).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product path (himg_amd/) never calls into it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks it byte-for-byte
 * against streams produced by the real reference compiled from
 * /root/reference (oracle/_ref, recipe in oracle/Makefile) and against the
 * golden values recorded in tests/golden/.
 */
#ifndef HIMG_ORACLE_H_
#define HIMG_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Intermediate products of one encode, for stage-by-stage GPU parity checks.
 * All buffers are owned by the trace and released by himg_oracle_trace_free. */
typedef struct {
  int width, height, channels, rows, cols, use_ycbcr;
  uint8_t *lifted;       /* W*H*stride bytes after the colour lift (ycbcr.cpp:24-52) */
  uint8_t *avg;          /* [C][rows][cols] box averages (downsampled.cpp:76-96) */
  uint8_t *lowres;       /* [C][rows][cols] phase-blended m_data (downsampled.cpp:98-113) */
  uint8_t *lres_sym;     /* [C][chan_size] predictor bytes + deltas (downsampled.cpp:177-316) */
  int lres_sym_size;
  uint8_t *fres_sym;     /* [rows][C][64][cols] companded coefficients (encoder.cpp:258-327) */
  int fres_sym_size;
  uint32_t lres_hist[261], fres_hist[261];   /* token histograms (huffman_enc.cpp:98-144) */
  uint8_t lres_len[261], fres_len[261];      /* code lengths */
  uint64_t lres_code[261], fres_code[261];   /* LSB-first codes */
  int lres_tree_bytes, fres_tree_bytes;
  int *fres_row_bytes;   /* [rows] payload bytes per block row */
  uint8_t shift_luma[64], shift_chroma[64];
  int16_t lmap[128], fmap[128];
} himg_oracle_trace;

/* Encode (encoder.cpp:59-109). Returns 0 on success; *out is malloc'ed. */
int himg_oracle_encode(const uint8_t *data, int width, int height,
                       int pixel_stride, int num_channels, int quality,
                       int use_ycbcr, uint8_t **out, int *out_size,
                       himg_oracle_trace *trace /* may be NULL */);

/* Decode (decoder.cpp:87-138). Returns 0 on success, a negative stage code on
 * the same inputs the reference rejects; *out is malloc'ed W*H*C bytes.
 * max_threads mirrors Decoder(int) (decoder.cpp:79-85); rows are decoded by
 * that many pthreads (<=0 -> number of online CPUs). */
int himg_oracle_decode(const uint8_t *packed, int packed_size, int max_threads,
                       uint8_t **out, int *width, int *height, int *channels);

/* Decode, also returning the entropy-decoded symbol planes (for GPU stage
 * checks).  lres_sym / fres_sym are malloc'ed, same layouts as the trace. */
int himg_oracle_decode_trace(const uint8_t *packed, int packed_size,
                             uint8_t **out, int *width, int *height,
                             int *channels, uint8_t **lres_sym,
                             int *lres_sym_size, uint8_t **fres_sym,
                             int *fres_sym_size, uint8_t **lowres);

/* Test knob for the product's opt-in fixed mode (trap T2); off = the reference. */
void himg_oracle_set_compat_fix(int on);

void himg_oracle_free(void *p);
void himg_oracle_trace_free(himg_oracle_trace *t);

/* Table builders, exposed for known-answer tests. */
void himg_oracle_shift_table(int quality, int chroma, uint8_t out[64]);
void himg_oracle_lowres_map_table(int quality, int16_t out[128]);
void himg_oracle_fullres_map_table(int16_t out[128]);
uint8_t himg_oracle_map_to_8bit(const int16_t table[128], int x);

/* Stage functions, exposed for unit tests. */
void himg_oracle_hadamard_forward(int16_t *out, const int16_t *in);
void himg_oracle_hadamard_inverse(int16_t *out, const int16_t *in);

#ifdef __cplusplus
}
#endif
#endif /* HIMG_ORACLE_H_ */
