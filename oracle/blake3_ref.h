/* ORACLE (test infrastructure, NOT product code): portable BLAKE3 (see blake3_ref.c). */
#ifndef ORACLE_BLAKE3_REF_H
#define ORACLE_BLAKE3_REF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* BLAKE3 default-mode hash, 32-byte output; stands in for blake3::hash() of the `blake3` crate
 * (/root/reference/crypto/src/hash/blake/mod.rs:27-29). */
void orc_blake3_hash(const uint8_t *in, size_t len, uint8_t out[32]);
#ifdef __cplusplus
}
#endif
#endif
