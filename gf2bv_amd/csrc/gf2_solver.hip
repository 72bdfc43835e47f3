// gf2_solver.hip -- host side of libgf2bv_hip.so: the C ABI declared in include/gf2bv_hip.h.
//
// This is what gf2bv/_internal.c:398-489 (assembly -> _mzd_pluq -> _mzd_pluq_solve_left ->
// _mzd_kernel_left_pluq -> transpose) becomes on an MI355X: one uninterrupted stream of HIP
// launches per system; every data-dependent quantity (rank so far, pivots of the current
// panel, row moves) stays in device memory, so the host never synchronises mid-elimination.
//
// There is deliberately NO CPU fallback: without a HIP device every solve entry point
// returns GF2BV_ERR_NODEVICE.
#include "gf2_kernels.hip.h"
#include "../../include/gf2bv_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err = "";

int fail(int code, const char *what, hipError_t e = hipSuccess)
{
	char buf[512];
	if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
	else snprintf(buf, sizeof buf, "%s", what);
	g_err = buf;
	return code;
}

#define HIPCHK(call)                                                     \
	do {                                                                 \
		hipError_t _e = (call);                                          \
		if (_e != hipSuccess) return fail(GF2BV_ERR_HIP, #call, _e);     \
	} while (0)

inline i64 round_up(i64 v, i64 m) { return (v + m - 1) / m * m; }

// ---- sweep configurations ------------------------------------------------------------------
struct SweepImpl {
	int K, TW, T, lds_bytes, threads;
	hipError_t (*sweep)(dim3, hipStream_t, u64 *, i64, i64, const PanelRec *, int, const u64 *, int, int, int);
	hipError_t (*apply)(dim3, hipStream_t, u64 *, i64, int, int, const SolveState *, const PanelRec *);
};

template <int K, int TW, int NT>
hipError_t launch_sweep(dim3 grid, hipStream_t s, u64 *M, i64 stride, i64 rows, const PanelRec *rec,
                        int above, const u64 *mult, int tile0, int ntiles, int rpb)
{
	static bool attr_set[16] = {};
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (dev < 16 && !attr_set[dev]) {
		constexpr int lds_attr = SweepCfg<K, TW>::LDS_BYTES;
		hipError_t e = hipFuncSetAttribute((const void *)k_sweep<K, TW, NT>,
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr);
		if (e != hipSuccess) return e;
		attr_set[dev] = true;
	}
	constexpr int lds = SweepCfg<K, TW>::LDS_BYTES;
	k_sweep<K, TW, NT><<<grid, dim3(NT), lds, s>>>(M, stride, rows, rec, above, mult, tile0, ntiles, rpb);
	return hipGetLastError();
}
template <int TW>
hipError_t launch_apply(dim3 grid, hipStream_t s, u64 *M, i64 stride, int j, int tile0,
                        const SolveState *st, const PanelRec *panels)
{
	hipLaunchKernelGGL((k_pivot_apply<TW>), grid, dim3(256), 0, s, M, stride, j, tile0, st, panels);
	return hipGetLastError();
}

#define SWEEP_IMPL(K, TW, NT) \
	{ K, TW, SweepCfg<K, TW>::T, SweepCfg<K, TW>::LDS_BYTES, NT, launch_sweep<K, TW, NT>, launch_apply<TW> }

const SweepImpl kImpls[] = {
	SWEEP_IMPL(7, 16, 1024),   // default: 10 tables, 128-byte row segments, 144 KiB LDS
	SWEEP_IMPL(8, 8, 1024),    // 8 tables, 64-byte segments, 128 KiB
	SWEEP_IMPL(6, 16, 1024),   // 11 tables, 128-byte segments, 82 KiB
	SWEEP_IMPL(5, 16, 1024),   // 13 tables, 128-byte segments, 50 KiB
	SWEEP_IMPL(5, 32, 1024),   // 13 tables, 256-byte segments, 100 KiB
	SWEEP_IMPL(4, 32, 1024),   // 16 tables, 256-byte segments, 64 KiB
	SWEEP_IMPL(6, 16, 512),
	SWEEP_IMPL(5, 16, 512),
};

const SweepImpl *pick_impl(i64 stride)
{
	const SweepImpl *chosen = &kImpls[0];
	if (const char *e = getenv("GF2BV_SWEEP")) {
		int k = 0, tw = 0, nt = 1024;
		if (sscanf(e, "%dx%dx%d", &k, &tw, &nt) >= 2)
			for (const SweepImpl &c : kImpls)
				if (c.K == k && c.TW == tw && c.threads == nt) { chosen = &c; break; }
	}
	if (stride % chosen->TW != 0) chosen = &kImpls[0];     // TW=16 always divides (stride % 16 == 0)
	return chosen;
}

// ---- one solve ---------------------------------------------------------------------------------
struct Solver {
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	u64 *M = nullptr;
	bool own_M = false;
	i64 rows = 0, cols = 0, stride = 0;
	int mode = 0;
	bool time_kernels = false;
	const SweepImpl *impl = nullptr;

	SolveState *st = nullptr;
	PanelRec *panels = nullptr;
	int *pivcol = nullptr;
	u64 *mult = nullptr;
	int *cand_cnt = nullptr, *cand_rows = nullptr;
	int units = 0;
	u64 *Y = nullptr;
	int *ycols = nullptr;
	u64 *out = nullptr;
	i64 ys = 0;
	int ny = 0;
	i64 maxr = 0;
	int npanels = 0;
	i64 wt = 0, cw = 0;
	int rpb = 2048;

	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
	std::vector<hipEvent_t> kev;
	std::vector<int> free_order;     // free columns in M4RI kernel order (mode 1)
	std::chrono::steady_clock::time_point t_begin;
	float ms_pack = 0;

	~Solver() { release(); }
	void release()
	{
		(void)hipSetDevice(device);
		for (void *p : { (void *)st, (void *)panels, (void *)pivcol, (void *)mult, (void *)cand_cnt,
		                 (void *)cand_rows, (void *)Y, (void *)ycols, (void *)out })
			if (p) (void)hipFree(p);
		st = nullptr; panels = nullptr; pivcol = nullptr; mult = nullptr; cand_cnt = nullptr;
		cand_rows = nullptr; Y = nullptr; ycols = nullptr; out = nullptr;
		if (own_M && M) (void)hipFree(M);
		M = nullptr;
		for (hipEvent_t e : { ev0, ev1, ev2, ev3 }) if (e) (void)hipEventDestroy(e);
		ev0 = ev1 = ev2 = ev3 = nullptr;
		for (hipEvent_t e : kev) (void)hipEventDestroy(e);
		kev.clear();
		if (own_stream && stream) (void)hipStreamDestroy(stream);
		stream = nullptr;
	}
};

}  // namespace

struct gf2bv_result {
	int status = 0;
	i64 rank = 0, dim = 0, cw = 0;
	std::vector<u64> origin, basis;
	std::vector<int32_t> pivots;
	gf2bv_stats stats{};
};

namespace {

int check_device(int device)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return fail(GF2BV_ERR_NODEVICE, "no HIP device visible: gf2bv_amd has no CPU fallback");
	if (device < 0 || device >= n) return fail(GF2BV_ERR_ARG, "device index out of range");
	HIPCHK(hipSetDevice(device));
	return GF2BV_OK;
}

int solver_alloc(Solver &S)
{
	S.wt = (S.cols + 1 + 63) / 64;
	S.cw = (S.cols + 63) / 64;
	S.npanels = (int)((S.cols + 63) / 64);
	S.maxr = std::min(S.rows, S.cols);
	S.impl = pick_impl(S.stride);
	// scan units: enough wavefronts to cover the rows 256 at a time, at most 256 units
	i64 want = (S.rows + 1023) / 1024;
	int nA = (int)std::min<i64>(64, std::max<i64>(1, (want + 3) / 4));
	S.units = nA * 4;
	i64 rpb = 2048;
	// keep >= ~2k workgroups in a full sweep when the matrix allows it
	{
		i64 ntiles = std::max<i64>(1, S.wt / S.impl->TW);
		while (rpb > 512 && ntiles * ((S.rows + rpb - 1) / rpb) < 2048) rpb /= 2;
		if (const char *e = getenv("GF2BV_RPB")) { int v = atoi(e); if (v >= 64) rpb = v; }
	}
	S.rpb = (int)rpb;
	HIPCHK(hipMalloc(&S.st, sizeof(SolveState)));
	HIPCHK(hipMalloc(&S.panels, sizeof(PanelRec) * std::max(1, S.npanels)));
	HIPCHK(hipMalloc(&S.pivcol, sizeof(int) * std::max<i64>(1, S.maxr + 64)));
	HIPCHK(hipMalloc(&S.mult, sizeof(u64) * std::max<i64>(1, S.rows)));
	HIPCHK(hipMalloc(&S.cand_cnt, sizeof(int) * S.units));
	HIPCHK(hipMalloc(&S.cand_rows, sizeof(int) * S.units * 64));
	HIPCHK(hipMemsetAsync(S.st, 0, sizeof(SolveState), S.stream));
	HIPCHK(hipMemsetAsync(S.panels, 0, sizeof(PanelRec) * std::max(1, S.npanels), S.stream));
	HIPCHK(hipEventCreate(&S.ev0));
	HIPCHK(hipEventCreate(&S.ev1));
	HIPCHK(hipEventCreate(&S.ev2));
	HIPCHK(hipEventCreate(&S.ev3));
	return GF2BV_OK;
}

// forward elimination: all panels, no host synchronisation
int enqueue_forward(Solver &S)
{
	const SweepImpl &I = *S.impl;
	const int TW = I.TW;
	const int tiles_total = (int)((S.wt + TW - 1) / TW);
	const i64 nrb = (S.rows + S.rpb - 1) / S.rpb;
	HIPCHK(hipEventRecord(S.ev0, S.stream));
	for (int j = 0; j < S.npanels; j++) {
		const i64 c0 = (i64)j * 64;
		const u64 colmask = (S.cols - c0 >= 64) ? ~0ull : ((1ull << (S.cols - c0)) - 1);
		const int tile0 = j / TW;
		const int ntiles = tiles_total - tile0;
		hipLaunchKernelGGL(k_panel_scan, dim3(S.units / 4), dim3(256), 0, S.stream,
		                   S.M, S.stride, S.rows, j, colmask, S.st, S.cand_cnt, S.cand_rows, S.units);
		hipLaunchKernelGGL(k_panel_select, dim3(1), dim3(64), 0, S.stream,
		                   S.M, S.stride, S.rows, j, colmask, S.st, S.panels, S.pivcol,
		                   S.cand_cnt, S.cand_rows, S.units);
		HIPCHK(I.apply(dim3(ntiles), S.stream, S.M, S.stride, j, tile0, S.st, S.panels));
		{
			int g = (int)std::min<i64>(1024, (S.rows + 255) / 256);
			hipLaunchKernelGGL(k_gather_mult, dim3(g), dim3(256), 0, S.stream,
			                   S.M, S.stride, S.rows, j, S.panels + j, 0, S.mult);
		}
		if (S.time_kernels) {
			hipEvent_t a, b;
			HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
			S.kev.push_back(a); S.kev.push_back(b);
			HIPCHK(hipEventRecord(a, S.stream));
		}
		HIPCHK(I.sweep(dim3((unsigned)(ntiles * nrb)), S.stream, S.M, S.stride, S.rows, S.panels + j, 0,
		               S.mult, tile0, ntiles, S.rpb));
		if (S.time_kernels) HIPCHK(hipEventRecord(S.kev.back(), S.stream));
	}
	{
		int g = (int)std::min<i64>(1024, (S.rows + 255) / 256);
		hipLaunchKernelGGL(k_check_rhs, dim3(g), dim3(256), 0, S.stream, S.M, S.stride, S.rows, S.cols, S.st);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(S.ev1, S.stream));
	return GF2BV_OK;
}

// back-substitution on Y = selected columns of U (RHS [+ free columns]); then scatter.
int enqueue_backward(Solver &S, const std::vector<int> &ycols_host)
{
	const SweepImpl &I = *S.impl;
	const int TW = I.TW;
	S.ny = (int)ycols_host.size();
	const i64 nyw = (S.ny + 63) / 64;
	S.ys = round_up(nyw, TW);
	HIPCHK(hipMalloc(&S.ycols, sizeof(int) * S.ny));
	HIPCHK(hipMemcpyAsync(S.ycols, ycols_host.data(), sizeof(int) * S.ny, hipMemcpyHostToDevice, S.stream));
	HIPCHK(hipMalloc(&S.Y, sizeof(u64) * std::max<i64>(1, S.maxr) * S.ys));
	HIPCHK(hipMemsetAsync(S.Y, 0, sizeof(u64) * std::max<i64>(1, S.maxr) * S.ys, S.stream));
	HIPCHK(hipMalloc(&S.out, sizeof(u64) * S.ny * std::max<i64>(1, S.cw)));
	HIPCHK(hipMemsetAsync(S.out, 0, sizeof(u64) * S.ny * std::max<i64>(1, S.cw), S.stream));
	if (S.maxr > 0) {
		i64 waves = S.maxr * nyw;
		hipLaunchKernelGGL(k_extract_y, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, S.stream,
		                   S.M, S.stride, S.st, S.ycols, S.ny, S.Y, S.ys);
		const int ytiles = (int)(S.ys / TW);
		for (int q = S.npanels - 1; q >= 1; q--) {
			// rows above panel q's pivots: at most min(64*q, maxr)
			const i64 bound = std::min<i64>((i64)64 * q, S.maxr);
			int g = (int)std::min<i64>(1024, (bound + 255) / 256);
			hipLaunchKernelGGL(k_gather_mult, dim3(g), dim3(256), 0, S.stream,
			                   S.M, S.stride, S.rows, q, S.panels + q, 1, S.mult);
			const i64 nrb = (bound + S.rpb - 1) / S.rpb;
			HIPCHK(I.sweep(dim3((unsigned)(ytiles * nrb)), S.stream, S.Y, S.ys, S.maxr, S.panels + q, 1,
			               S.mult, 0, ytiles, S.rpb));
		}
		i64 thr = S.maxr * nyw;
		hipLaunchKernelGGL(k_scatter_solution, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, S.stream,
		                   S.Y, S.ys, S.st, S.pivcol, S.ny, S.out, std::max<i64>(1, S.cw));
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(S.ev2, S.stream));
	return GF2BV_OK;
}

int solver_enqueue(Solver &S)
{
	int rc = solver_alloc(S);
	if (rc) return rc;
	rc = enqueue_forward(S);
	if (rc) return rc;
	if (S.mode == GF2BV_MODE_SINGLE) {
		std::vector<int> yc(1, (int)S.cols);
		return enqueue_backward(S, yc);
	}
	return GF2BV_OK;
}

int solver_finish(Solver &S, gf2bv_result **out)
{
	SolveState hst;
	std::vector<int32_t> piv;
	if (S.mode == GF2BV_MODE_AFFINE_SPACE) {
		// the kernel basis needs rank and pivot columns on the host (one sync) to lay out
		// the free columns in M4RI's order (_internal.c:348)
		HIPCHK(hipMemcpyAsync(&hst, S.st, sizeof hst, hipMemcpyDeviceToHost, S.stream));
		HIPCHK(hipStreamSynchronize(S.stream));
		piv.resize(hst.rank);
		if (hst.rank)
			HIPCHK(hipMemcpy(piv.data(), S.pivcol, sizeof(int) * hst.rank, hipMemcpyDeviceToHost));
		std::vector<int> order(S.cols);
		for (i64 i = 0; i < S.cols; i++) order[i] = (int)i;
		for (int i = 0; i < hst.rank; i++) std::swap(order[i], order[piv[i]]);
		S.free_order.assign(order.begin() + hst.rank, order.end());
		std::vector<int> yc;
		if (!hst.inconsistent) yc = S.free_order;
		yc.push_back((int)S.cols);
		int rc = enqueue_backward(S, yc);
		if (rc) return rc;
	}
	HIPCHK(hipMemcpyAsync(&hst, S.st, sizeof hst, hipMemcpyDeviceToHost, S.stream));
	std::vector<u64> hout((size_t)S.ny * std::max<i64>(1, S.cw));
	HIPCHK(hipMemcpyAsync(hout.data(), S.out, sizeof(u64) * hout.size(), hipMemcpyDeviceToHost, S.stream));
	std::vector<PanelRec> hp(std::max(1, S.npanels));
	HIPCHK(hipMemcpyAsync(hp.data(), S.panels, sizeof(PanelRec) * hp.size(), hipMemcpyDeviceToHost, S.stream));
	HIPCHK(hipEventRecord(S.ev3, S.stream));
	HIPCHK(hipStreamSynchronize(S.stream));
	if (piv.empty() && hst.rank) {
		piv.resize(hst.rank);
		HIPCHK(hipMemcpy(piv.data(), S.pivcol, sizeof(int) * hst.rank, hipMemcpyDeviceToHost));
	}

	gf2bv_result *R = new gf2bv_result();
	R->status = hst.inconsistent ? GF2BV_STATUS_INCONSISTENT : GF2BV_STATUS_SOLVED;
	R->rank = hst.rank;
	R->cw = S.cw;
	R->dim = S.cols - hst.rank;
	R->pivots = piv;
	R->origin.assign(std::max<i64>(1, S.cw), 0);
	if (R->status == GF2BV_STATUS_SOLVED) {
		const u64 *o = hout.data() + (size_t)(S.ny - 1) * std::max<i64>(1, S.cw);
		std::copy(o, o + S.cw, R->origin.begin());
		if (S.mode == GF2BV_MODE_AFFINE_SPACE) {
			R->basis.assign((size_t)R->dim * std::max<i64>(1, S.cw), 0);
			for (i64 t = 0; t < R->dim; t++) {
				u64 *v = R->basis.data() + (size_t)t * S.cw;
				std::copy(hout.data() + (size_t)t * S.cw, hout.data() + (size_t)(t + 1) * S.cw, v);
				int f = S.free_order[t];
				v[f >> 6] |= 1ull << (f & 63);
			}
		}
	}
	gf2bv_stats &st = R->stats;
	st.rows = S.rows; st.cols = S.cols; st.stride_words = S.stride;
	st.rank = R->rank; st.dimension = R->dim; st.status = R->status;
	st.n_panels = S.npanels;
	st.tables_per_sweep = S.impl->T; st.table_bits = S.impl->K; st.tile_words = S.impl->TW;
	for (int j = 0; j < S.npanels; j++) {
		if (hp[j].p <= 0) continue;
		double rows_swept = (double)(S.rows - hp[j].start - hp[j].p);
		st.n_sweeps++;
		st.sweep_words += rows_swept * (double)(S.wt - j);
		st.row_xors += rows_swept * (double)S.impl->T;
	}
	st.ms_pack = S.ms_pack;
	(void)hipEventElapsedTime(&st.ms_eliminate, S.ev0, S.ev1);
	(void)hipEventElapsedTime(&st.ms_backsub, S.ev1, S.ev2);
	(void)hipEventElapsedTime(&st.ms_export, S.ev2, S.ev3);
	for (size_t i = 0; i + 1 < S.kev.size(); i += 2) {
		float ms = 0;
		(void)hipEventElapsedTime(&ms, S.kev[i], S.kev[i + 1]);
		st.ms_sweep += ms;
	}
	st.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - S.t_begin).count();
	*out = R;
	return GF2BV_OK;
}

int check_shape(i64 rows, i64 cols, int mode)
{
	// mirrors gf2bv/_internal.c:372-395
	if (cols <= 0) return fail(GF2BV_ERR_ARG, "Number of columns must be positive");
	if (mode != GF2BV_MODE_SINGLE && mode != GF2BV_MODE_AFFINE_SPACE) return fail(GF2BV_ERR_ARG, "Invalid mode");
	if (rows < cols)
		return fail(GF2BV_ERR_ARG, "Number of rows must be greater than or equal to number of columns, try pad with zeros.");
	if (rows >= (1ll << 31) - 64 || cols >= (1ll << 31) - 64) return fail(GF2BV_ERR_ARG, "system too large");
	return GF2BV_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int gf2bv_version(void) { return 100; }

int gf2bv_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

const char *gf2bv_last_error(void) { return g_err.c_str(); }

int gf2bv_solve_device(void *d_aug, int64_t rows, int64_t cols, int64_t stride_words, int mode,
                       int device, void *stream, int time_kernels, gf2bv_result **out)
{
	if (!out || !d_aug) return fail(GF2BV_ERR_ARG, "null pointer");
	*out = nullptr;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	if (stride_words % 16 != 0 || stride_words < (cols + 1 + 63) / 64 || ((uintptr_t)d_aug & 15))
		return fail(GF2BV_ERR_ARG, "device matrix needs 16-byte alignment and stride_words % 16 == 0 covering cols+1 bits");
	rc = check_device(device);
	if (rc) return rc;
	Solver S;
	S.t_begin = std::chrono::steady_clock::now();
	S.device = device;
	S.stream = (hipStream_t)stream;
	S.M = (u64 *)d_aug;
	S.rows = rows; S.cols = cols; S.stride = stride_words; S.mode = mode;
	S.time_kernels = time_kernels != 0;
	rc = solver_enqueue(S);
	if (rc) return rc;
	return solver_finish(S, out);
}

int gf2bv_solve_batch_device(void *d_aug, int64_t nsys, int64_t sys_stride_words, int64_t rows, int64_t cols,
                             int64_t stride_words, int mode, int device, gf2bv_result **out)
{
	if (!out || !d_aug || nsys < 0) return fail(GF2BV_ERR_ARG, "null pointer");
	for (i64 s = 0; s < nsys; s++) out[s] = nullptr;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	if (stride_words % 16 != 0 || stride_words < (cols + 1 + 63) / 64 || sys_stride_words < rows * stride_words ||
	    ((uintptr_t)d_aug & 15) || (sys_stride_words & 1))
		return fail(GF2BV_ERR_ARG, "bad batch layout");
	rc = check_device(device);
	if (rc) return rc;
	// independent systems: a few in flight on their own streams so one system's
	// latency-bound panel steps overlap another system's sweeps
	const int NS = (int)std::min<i64>(nsys, 4);
	std::vector<hipStream_t> streams(NS);
	for (int i = 0; i < NS; i++) HIPCHK(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
	int result = GF2BV_OK;
	for (i64 s0 = 0; s0 < nsys && result == GF2BV_OK; s0 += NS) {
		const int nb = (int)std::min<i64>(NS, nsys - s0);
		std::vector<Solver> group(nb);
		for (int i = 0; i < nb && result == GF2BV_OK; i++) {
			Solver &S = group[i];
			S.t_begin = std::chrono::steady_clock::now();
			S.device = device;
			S.stream = streams[i];
			S.M = (u64 *)d_aug + (s0 + i) * sys_stride_words;
			S.rows = rows; S.cols = cols; S.stride = stride_words; S.mode = mode;
			result = solver_enqueue(S);
		}
		for (int i = 0; i < nb && result == GF2BV_OK; i++) result = solver_finish(group[i], &out[s0 + i]);
		for (int i = 0; i < nb; i++) (void)hipStreamSynchronize(streams[i]);
	}
	for (int i = 0; i < NS; i++) (void)hipStreamDestroy(streams[i]);
	return result;
}

int gf2bv_solve_words(const uint64_t *aug, int64_t rows, int64_t cols, int64_t stride_words, int mode,
                      int device, gf2bv_result **out)
{
	if (!out || (!aug && rows > 0)) return fail(GF2BV_ERR_ARG, "null pointer");
	*out = nullptr;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	const i64 wt = (cols + 1 + 63) / 64;
	if (stride_words < wt) return fail(GF2BV_ERR_ARG, "stride_words does not cover cols+1 bits");
	rc = check_device(device);
	if (rc) return rc;
	Solver S;
	S.t_begin = std::chrono::steady_clock::now();
	S.device = device;
	HIPCHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
	S.own_stream = true;
	S.rows = rows; S.cols = cols; S.mode = mode;
	S.stride = round_up(wt, 32);
	HIPCHK(hipMalloc(&S.M, sizeof(u64) * std::max<i64>(1, rows) * S.stride));
	S.own_M = true;
	hipEvent_t p0, p1;
	HIPCHK(hipEventCreate(&p0)); HIPCHK(hipEventCreate(&p1));
	HIPCHK(hipEventRecord(p0, S.stream));
	HIPCHK(hipMemsetAsync(S.M, 0, sizeof(u64) * std::max<i64>(1, rows) * S.stride, S.stream));
	if (rows > 0)
		HIPCHK(hipMemcpy2DAsync(S.M, S.stride * 8, aug, stride_words * 8, wt * 8, rows, hipMemcpyHostToDevice, S.stream));
	// bits above column `cols` are ignored by the reference (_internal.c:414): they can only sit
	// in the last data word and are never used as pivots (colmask) nor exported; the RHS bit is
	// read at exactly column `cols`.  A stray high bit could still leak through XORs into rows'
	// tails, which nobody reads.  Nothing to mask.
	HIPCHK(hipEventRecord(p1, S.stream));
	rc = solver_enqueue(S);
	if (rc == GF2BV_OK) {
		(void)hipEventSynchronize(p1);
		(void)hipEventElapsedTime(&S.ms_pack, p0, p1);
		rc = solver_finish(S, out);
	}
	(void)hipEventDestroy(p0); (void)hipEventDestroy(p1);
	return rc;
}

int gf2bv_solve_digits(const uint32_t *digits, const int64_t *digit_off, int bits_per_digit, int64_t rows,
                       int64_t cols, int mode, int device, gf2bv_result **out)
{
	if (!out || !digit_off) return fail(GF2BV_ERR_ARG, "null pointer");
	*out = nullptr;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	if (bits_per_digit < 1 || bits_per_digit > 32) return fail(GF2BV_ERR_ARG, "bits_per_digit must be 1..32");
	rc = check_device(device);
	if (rc) return rc;
	Solver S;
	S.t_begin = std::chrono::steady_clock::now();
	S.device = device;
	HIPCHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
	S.own_stream = true;
	S.rows = rows; S.cols = cols; S.mode = mode;
	const i64 wt = (cols + 1 + 63) / 64;
	S.stride = round_up(wt, 32);
	HIPCHK(hipMalloc(&S.M, sizeof(u64) * std::max<i64>(1, rows) * S.stride));
	S.own_M = true;
	const i64 ndig = digit_off[rows];
	uint32_t *d_dig = nullptr;
	i64 *d_off = nullptr;
	HIPCHK(hipMalloc(&d_dig, sizeof(uint32_t) * std::max<i64>(1, ndig)));
	HIPCHK(hipMalloc(&d_off, sizeof(i64) * (rows + 1)));
	hipEvent_t p0, p1;
	HIPCHK(hipEventCreate(&p0)); HIPCHK(hipEventCreate(&p1));
	HIPCHK(hipEventRecord(p0, S.stream));
	if (ndig) HIPCHK(hipMemcpyAsync(d_dig, digits, sizeof(uint32_t) * ndig, hipMemcpyHostToDevice, S.stream));
	HIPCHK(hipMemcpyAsync(d_off, digit_off, sizeof(i64) * (rows + 1), hipMemcpyHostToDevice, S.stream));
	{
		i64 total = rows * S.stride;
		hipLaunchKernelGGL(k_pack_digits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S.stream,
		                   d_dig, d_off, bits_per_digit, (i64)rows, (i64)cols, S.stride, S.M);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(p1, S.stream));
	rc = solver_enqueue(S);
	if (rc == GF2BV_OK) {
		(void)hipEventSynchronize(p1);
		(void)hipEventElapsedTime(&S.ms_pack, p0, p1);
		rc = solver_finish(S, out);
	}
	(void)hipStreamSynchronize(S.stream);
	(void)hipFree(d_dig); (void)hipFree(d_off);
	(void)hipEventDestroy(p0); (void)hipEventDestroy(p1);
	return rc;
}

// ---- result accessors ----------------------------------------------------------------------------
int gf2bv_result_status(const gf2bv_result *r) { return r ? r->status : -1; }
int64_t gf2bv_result_rank(const gf2bv_result *r) { return r ? r->rank : -1; }
int64_t gf2bv_result_dimension(const gf2bv_result *r) { return r ? r->dim : -1; }
int64_t gf2bv_result_words(const gf2bv_result *r) { return r ? r->cw : -1; }
int gf2bv_result_origin(const gf2bv_result *r, uint64_t *o)
{
	if (!r || !o) return fail(GF2BV_ERR_ARG, "null pointer");
	memcpy(o, r->origin.data(), sizeof(u64) * r->cw);
	return GF2BV_OK;
}
int gf2bv_result_basis(const gf2bv_result *r, uint64_t *o)
{
	if (!r || (!o && !r->basis.empty())) return fail(GF2BV_ERR_ARG, "null pointer");
	if (!r->basis.empty()) memcpy(o, r->basis.data(), sizeof(u64) * r->basis.size());
	return GF2BV_OK;
}
int gf2bv_result_pivots(const gf2bv_result *r, int32_t *o)
{
	if (!r || (!o && !r->pivots.empty())) return fail(GF2BV_ERR_ARG, "null pointer");
	if (!r->pivots.empty()) memcpy(o, r->pivots.data(), sizeof(int32_t) * r->pivots.size());
	return GF2BV_OK;
}
int gf2bv_result_stats(const gf2bv_result *r, gf2bv_stats *o)
{
	if (!r || !o) return fail(GF2BV_ERR_ARG, "null pointer");
	*o = r->stats;
	return GF2BV_OK;
}
void gf2bv_result_free(gf2bv_result *r) { delete r; }

void gf2bv_space_combine(const uint64_t *origin, const uint64_t *basis, int64_t dimension, int64_t words,
                         const uint64_t *selector, int64_t selector_words, uint64_t *o)
{
	for (i64 w = 0; w < words; w++) o[w] = origin[w];
	for (i64 i = 0; i < dimension && (i >> 6) < selector_words; i++)
		if ((selector[i >> 6] >> (i & 63)) & 1)
			for (i64 w = 0; w < words; w++) o[w] ^= basis[i * words + w];
}

// ---- synthetic + residual + buffers --------------------------------------------------------------
int gf2bv_synth_device(void *d_aug, int64_t rows, int64_t cols, int64_t stride_words, uint64_t seed,
                       int device, void *stream)
{
	if (!d_aug || rows < 0 || cols <= 0 || stride_words < (cols + 1 + 63) / 64 || rows >= (1ll << 20) - 1)
		return fail(GF2BV_ERR_ARG, "bad synthetic shape");
	int rc = check_device(device);
	if (rc) return rc;
	if (rows == 0) return GF2BV_OK;
	hipLaunchKernelGGL(k_synth, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
	                   (u64 *)d_aug, (i64)rows, (i64)cols, (i64)stride_words, (u64)seed);
	HIPCHK(hipGetLastError());
	return GF2BV_OK;
}

int gf2bv_residual_device(const void *d_aug, int64_t rows, int64_t cols, int64_t stride_words,
                          const uint64_t *x_words, int device, void *stream, int64_t *bad_rows)
{
	if (!d_aug || !x_words || !bad_rows || cols <= 0) return fail(GF2BV_ERR_ARG, "null pointer");
	int rc = check_device(device);
	if (rc) return rc;
	const i64 cw = (cols + 63) / 64;
	u64 *dx = nullptr, *dbad = nullptr;
	HIPCHK(hipMalloc(&dx, sizeof(u64) * cw));
	HIPCHK(hipMalloc(&dbad, sizeof(u64)));
	hipStream_t s = (hipStream_t)stream;
	HIPCHK(hipMemcpyAsync(dx, x_words, sizeof(u64) * cw, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemsetAsync(dbad, 0, sizeof(u64), s));
	if (rows > 0)
		hipLaunchKernelGGL(k_residual, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
		                   (const u64 *)d_aug, (i64)rows, (i64)cols, (i64)stride_words, dx, dbad);
	u64 h = 0;
	HIPCHK(hipMemcpyAsync(&h, dbad, sizeof h, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	(void)hipFree(dx); (void)hipFree(dbad);
	*bad_rows = (int64_t)h;
	return GF2BV_OK;
}

int gf2bv_device_alloc(int device, int64_t bytes, void **d_ptr)
{
	if (!d_ptr || bytes < 0) return fail(GF2BV_ERR_ARG, "bad alloc request");
	int rc = check_device(device);
	if (rc) return rc;
	hipError_t e = hipMalloc(d_ptr, (size_t)std::max<i64>(bytes, 16));
	if (e != hipSuccess) return fail(GF2BV_ERR_NOMEM, "hipMalloc", e);
	return GF2BV_OK;
}
int gf2bv_device_free(int device, void *d_ptr)
{
	int rc = check_device(device);
	if (rc) return rc;
	HIPCHK(hipFree(d_ptr));
	return GF2BV_OK;
}
int gf2bv_device_upload(int device, void *d_dst, const void *h_src, int64_t bytes)
{
	int rc = check_device(device);
	if (rc) return rc;
	HIPCHK(hipMemcpy(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice));
	return GF2BV_OK;
}
int gf2bv_device_download(int device, void *h_dst, const void *d_src, int64_t bytes)
{
	int rc = check_device(device);
	if (rc) return rc;
	HIPCHK(hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
	return GF2BV_OK;
}

}  // extern "C"
