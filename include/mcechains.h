/* mcechains.h -- C ABI of libmcechains.so: multi-threaded reader for CosmoMC / MontePython chain
 * text files (host-only C++, no GPU, no ROCm dependency).
 *
 * Replaces `np.loadtxt(f)` at reference MCEvidence.py:564 (`read_list_to_array`), which costs ~10 s
 * per 1M x 29 text rows -- comparable to, or larger than, the whole GPU evidence computation
 * (SURVEY.md section 8f.4).  Same semantics as that call on such files:
 *   - fields separated by ASCII whitespace; '#' starts a comment that runs to the end of the line;
 *     blank / comment-only lines are skipped; lines end at "\n", "\r\n" or a bare "\r";
 *   - every field is parsed to the correctly rounded fp64 value (bit-identical to Python's float());
 *     "inf", "nan", "infinity" (any case, optional sign) are accepted;
 *   - all data lines must have the same number of fields (np.loadtxt raises ValueError otherwise).
 * Burn-in, thinning and concatenation stay in Python (mcevidence_amd/chains.py).
 *
 * Protocol: open (mmap + count rows/columns) -> read into a caller-owned row-major buffer -> close.
 * Return 0 on success, negative MCC_ERR_* otherwise; mce_chain_last_error() gives a thread-local
 * message.  Handles are not shared between threads. */
#ifndef MCECHAINS_H
#define MCECHAINS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCC_OK 0
#define MCC_ERR_IO (-1)      /* cannot open / stat / map the file            -> OSError    */
#define MCC_ERR_PARSE (-2)   /* a field is not a number                      -> ValueError */
#define MCC_ERR_RAGGED (-3)  /* the number of columns changed                -> ValueError */
#define MCC_ERR_INVALID (-4) /* bad argument                                 -> ValueError */

int mce_chain_abi_version(void);
const char *mce_chain_last_error(void);

/* nthreads <= 0: one thread per ~4 MB of file, at most the hardware concurrency (capped at 32). */
int mce_chain_open(const char *path, int32_t nthreads, void **handle, int64_t *nrows, int64_t *ncols);
int mce_chain_read(void *handle, double *out /* [nrows * ncols], row-major */);
void mce_chain_close(void *handle);

/* Parses one numeric token (no surrounding whitespace) exactly as the reader does; for tests. */
int mce_chain_parse_token(const char *token, int64_t len, double *value);

/* 64-bit fingerprint of the dense array rows[n][d] (row stride ld doubles) on the host's cores: sum over the words i = r d + c
 * of mix64(word_i + (salt + i) * 0x9E3779B97F4A7C15) -- bit for bit what libmcevidence_hip's device-side checksum computes over an
 * uploaded copy.  Multi-rank evidence() with one upload of the chain per node (mcevidence_amd/parallel.py) compares the
 * ranks' host fingerprints with it.  nthreads <= 0: one thread per 2 MB, at most the hardware concurrency (capped at 32). */
uint64_t mce_chain_fingerprint_f64(const double *rows, int64_t n, int64_t d, int64_t ld, uint64_t salt, int32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif /* MCECHAINS_H */
