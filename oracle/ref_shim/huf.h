/*
 * huf.h shim — TEST INFRASTRUCTURE ONLY.
 *
 * The reference's csrc/zipnn_core.c:10 does `#include "huf.h"` from an un-vendored
 * submodule (include/FiniteStateEntropy, empty in /root/reference).  The three entry
 * points it uses (zipnn_core.c:366,807,813) are exported, with the same signatures,
 * by the system libzstd.so.1 (zstd 1.4.8).  This shim only declares them so that the
 * reference sources can be compiled from where they lie into oracle/_ref/.
 */
#ifndef ZN_REF_SHIM_HUF_H
#define ZN_REF_SHIM_HUF_H
#include <stddef.h>
size_t HUF_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);
size_t HUF_decompress(void* dst, size_t originalSize, const void* cSrc, size_t cSrcSize);
unsigned HUF_isError(size_t code);
#endif
