/* m4ri_timing.c -- OPTIONAL true-M4RI timing beside the CPU baseline (test infrastructure; bench.py's cpu_baseline leg only).
 *
 * The reference's solve is M4RI's: _mzd_pluq(A, P, Q, 0) + _mzd_pluq_solve_left(A, r, P, Q, B, 0, 1) (gf2bv/_internal.c:431-440).
 * M4RI (release 20260122, setup.py:14-17) is neither vendored in /root/reference nor installed in this image, so the oracle is a
 * restatement (gf2_oracle.c) and the CPU baseline is that restatement (`kind: "port"`).  On a box that DOES have libm4ri.so this file
 * times the real thing: the library is opened at run time (dlopen -- nothing is linked, no M4RI header is needed or imitated: only
 * exported functions with scalar / opaque-pointer signatures are called), an N x N matrix is filled by mzd_randomize (M4RI's own
 * generator: a dense random matrix of the benchmark's shape; the synthetic generator's bits cannot be written without the header's
 * inline accessors), and mzd_pluq + mzd_pluq_solve_left -- the public wrappers of the two calls above -- are timed.
 *   returns 0 and fills seconds / rank when the library was found and ran, 1 when there is no libm4ri (the normal case here),
 *   2 when a symbol is missing.
 * Not a parity pin: the input differs from the GPU run's.  (Pinning the rank-deficient tie-breaking against a live M4RI needs its
 * headers, i.e. a real oracle/_ref build: see oracle/Makefile `ref`.) */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>

typedef void *(*init_fn)(int, int);
typedef void (*free_fn)(void *);
typedef void *(*mzp_init_fn)(int);
typedef int (*pluq_fn)(void *, void *, void *, int);
typedef int (*solve_fn)(void const *, int, void const *, void const *, void *, int, int);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int gf2o_m4ri_time(int n, double *seconds_pluq, double *seconds_solve, int *rank, char *libname, int libname_len)
{
	static const char *names[] = { "libm4ri.so", "libm4ri.so.2", "libm4ri.so.1", "libm4ri-0.0.20200125.so", 0 };
	void *h = 0;
	for (int i = 0; names[i] && !h; i++) { h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL); if (h && libname) snprintf(libname, libname_len, "%s", names[i]); }
	if (!h) return 1;
	init_fn mzd_init = (init_fn)dlsym(h, "mzd_init");
	free_fn mzd_free = (free_fn)dlsym(h, "mzd_free"), mzd_randomize = (free_fn)dlsym(h, "mzd_randomize"), mzp_free = (free_fn)dlsym(h, "mzp_free");
	mzp_init_fn mzp_init = (mzp_init_fn)dlsym(h, "mzp_init");
	pluq_fn mzd_pluq = (pluq_fn)dlsym(h, "mzd_pluq");
	solve_fn mzd_pluq_solve_left = (solve_fn)dlsym(h, "mzd_pluq_solve_left");
	if (!mzd_init || !mzd_free || !mzd_randomize || !mzp_init || !mzp_free || !mzd_pluq || !mzd_pluq_solve_left) { dlclose(h); return 2; }
	void *A = mzd_init(n, n), *B = mzd_init(n, 1), *P = mzp_init(n), *Q = mzp_init(n);
	mzd_randomize(A); mzd_randomize(B);
	double t0 = now();
	const int r = mzd_pluq(A, P, Q, 0);
	double t1 = now();
	(void)mzd_pluq_solve_left(A, r, P, Q, B, 0, 1);
	double t2 = now();
	*seconds_pluq = t1 - t0; *seconds_solve = t2 - t1; *rank = r;
	mzd_free(A); mzd_free(B); mzp_free(P); mzp_free(Q);
	dlclose(h);
	return 0;
}
