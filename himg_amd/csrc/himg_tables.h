/* himg_tables.h -- internal host-side table helpers (see himg_tables.c). */
#ifndef HIMG_TABLES_H_
#define HIMG_TABLES_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int himg_tables_unmap(const int16_t table[128], uint8_t code);
/* Serialise / parse an LMAP/FMAP chunk body; returns bytes written / 1 on success. */
int himg_tables_mapping_function(const int16_t table[128], uint8_t *out);
int himg_tables_parse_mapping_function(int16_t table[128], const uint8_t *in, int size);
#ifdef __cplusplus
}
#endif
#endif
