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
#include <hip/hip_ext.h>
#include "../../include/gf2bv_hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
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

// Nothing may propagate through the C ABI: allocation failures of the host-side containers become GF2BV_ERR_NOMEM.
// GF2BV_RETRY_EVENTS (internal, never returned through the C ABI): a stream hand-over gate gave up on the device -- kernels
// of the two streams were not executing concurrently after all (a profiler attached since the probe, say).  The device
// has been marked accordingly (streams_run_concurrently) and the solve is repeated once, with the event hand-over: the
// input is never modified.
#define GF2BV_RETRY_EVENTS (-77)
thread_local int g_attempt = 0;      // attempt of the running C-ABI call (gf2bv_stats::handover_retries)
template <class F>
int guarded(F &&body)
{
	for (int attempt = 0;; attempt++) {
		g_attempt = attempt;
		try {
			const int rc = body();
			if (rc != GF2BV_RETRY_EVENTS) return rc;
			if (attempt) return fail(GF2BV_ERR_HIP, "a stream hand-over gate timed out on the device");
		}
		catch (const std::bad_alloc &) { return fail(GF2BV_ERR_NOMEM, "out of host memory"); }
		catch (const std::exception &e) { return fail(GF2BV_ERR_HIP, e.what()); }
	}
}

#define HIPCHK(call)                                                     \
	do {                                                                 \
		hipError_t _e = (call);                                          \
		if (_e != hipSuccess) return fail(GF2BV_ERR_HIP, #call, _e);     \
	} while (0)

inline i64 round_up(i64 v, i64 m) { return (v + m - 1) / m * m; }
// k_update16k does not clamp its row indices: a tile's last chunk (an item: up to GF2_WSEG x 1024 rows) may read that far past the tile
// (never stored); the working matrix and the multiplier sets get this much slack behind them
constexpr size_t kOuterSlackBytes = (size_t)GF2_WSEG * 1024 * 32;   // (the largest item of the outer pass's workgroup shapes: 12288 rows)

struct Trace {
	bool on = getenv("GF2BV_TRACE") != nullptr;
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	void mark(const char *what)
	{
		if (!on) return;
		auto t = std::chrono::steady_clock::now();
		fprintf(stderr, "[gf2bv trace] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count());
		t0 = t;
	}
};
// rows of one tile slab: the rows padded to a wavefront's 64, plus 64 more (>= 1 is needed: rows nobody owns take the outer
// pass's dummy stores).  A slab is then a whole number of KiB, and a wavefront's batch -- 64 consecutive rows of a 16-byte tile
// -- is eight whole 128-byte lines.  Rounds 1-2 padded by 2 rows ("skew neighbouring tiles across the channels": 32 bytes per
// tile with 16-byte tiles): every batch of every tile but the first then straddled NINE lines.  65536^2: 33.7 -> 32.5 ms with
// any multiple of 64 (64 ... 1024 alike: the channel hash does not care about the stride); other sizes 0-1 %
// (profiles/r03_slab_pad.txt; GF2BV_SLAB_PAD=2 restores the old layout).
inline i64 slab_pad() { static const i64 pad = getenv("GF2BV_SLAB_PAD") ? std::max<i64>(1, atol(getenv("GF2BV_SLAB_PAD"))) : 64; return pad; }
inline i64 slab_rows(i64 rows) { return round_up(std::max<i64>(rows, 1), 64) + slab_pad(); }
// column tiles of a row of wt words: whole ownership units of 8 words (one 64-byte tile / four 16-byte tiles)
inline i64 tiles_for(i64 wt) { return round_up(std::max<i64>(wt, 1), (i64)1 << GF2_OWN_LOG) / GF2_TW; }


// ---- resource pool -----------------------------------------------------------------------------
// hipMalloc / hipFree / stream and event creation cost 0.1-1 ms each; a small solve (a few hundred
// microseconds of GPU time) would spend 5+ ms in them.  Buffers up to 64 MiB (512 MiB in total per
// process), streams and events are therefore recycled across calls.  Thread-safe; everything handed
// out is idle (a solve releases its resources only after synchronising its streams).
// GF2BV_PLAIN=1 (round 6): the floor without heuristics on undocumented hardware behaviour -- no stream-pair probing (any idle
// low-priority stream), no XCD pinning of gangs (the plain (spans, systems) grid), hand-overs between the streams through events instead
// of counters in memory + k_gate, every block enqueued with both panel paths (no optimistic enqueue).  Same answers (the parity file runs
// under it), the speed it costs is reported by bench.py as `plain_ms_per_step`.  An explicit knob still wins over it.
bool plain_mode()
{
	const char *e = getenv("GF2BV_PLAIN");
	return e && atoi(e) != 0;
}

struct Pool {
	std::mutex mu;
	std::unordered_map<void *, std::pair<int, size_t>> live;          // ptr -> (device, bucket bytes)
	std::map<std::pair<int, size_t>, std::vector<void *>> free_bufs;  // (device, bucket) -> buffers
	size_t cached_bytes = 0;
	std::map<std::pair<int, int>, std::vector<hipEvent_t>> events;   // (device, timing?) -> events
	std::map<std::pair<int, int>, std::vector<hipStream_t>> streams;  // (device, low priority?) -> streams
	static constexpr size_t kMaxBucket = (size_t)64 << 20, kMaxCached = (size_t)512 << 20;
	// A few LARGE buffers per device (the working matrices of the last big solves) are kept as well: hipMalloc +
	// hipFree of 2-8 GiB cost 10-100 ms per call, a visible part of a 0.3-2 s solve.  Up to kMaxBig of them (round 4: the
	// batch entry runs two or three gangs at a time, each with a working buffer of 3-4 GiB -- with ONE kept buffer every other
	// gang paid an allocation and a free).  Round 5: what is kept is bounded by the DEVICE (a sixth of its memory, 48 GiB at
	// most; GF2BV_KEEP_BIG=0: nothing is kept), an allocation that fails frees everything idle on that device and is
	// repeated once, gf2bv_pool_trim() does the same on request, and no hipFree runs under the pool's mutex.
	struct Big { void *p = nullptr; size_t bytes = 0; };
	static constexpr size_t kMaxBig = 6, kMaxBigBytes = (size_t)48 << 30;      // (a gang holds two: working matrices and the side-array arena)
	std::map<int, std::vector<Big>> big_free;                         // device -> idle large buffers
	std::unordered_map<void *, std::pair<int, size_t>> big_live;      // handed-out large buffers
	std::map<int, size_t> big_cap;                                    // device -> bytes of idle large buffers kept there
	static bool keep_big()
	{
		static const bool keep = !(getenv("GF2BV_KEEP_BIG") && atoi(getenv("GF2BV_KEEP_BIG")) == 0);
		return keep;
	}
	// (mu held) idle large buffers of `device` may hold this many bytes: a sixth of the device's memory, 48 GiB at most
	size_t big_cap_locked(int device)
	{
		auto it = big_cap.find(device);
		if (it != big_cap.end()) return it->second;
		size_t cap = kMaxBigBytes, fr = 0, total = 0;
		int cur = -1;
		(void)hipGetDevice(&cur);
		if (cur == device && hipMemGetInfo(&fr, &total) == hipSuccess && total) cap = std::min(cap, total / 6);
		else if (cur != device) return cap;       // (not the calling thread's device: ask again later)
		big_cap[device] = cap;
		return cap;
	}

	static size_t bucket(size_t bytes)
	{
		size_t b = 4096;
		while (b < bytes) b <<= 1;
		return b;
	}
	// every idle buffer of `device` (large ones and the bucket cache) leaves the pool; the caller frees them OUTSIDE the lock
	void take_idle_locked(int device, std::vector<void *> &drop)
	{
		auto bf = big_free.find(device);
		if (bf != big_free.end()) { for (const Big &b : bf->second) drop.push_back(b.p); bf->second.clear(); }
		for (auto &kv : free_bufs)
			if (kv.first.first == device) {
				for (void *q : kv.second) { drop.push_back(q); cached_bytes -= kv.first.second; }
				kv.second.clear();
			}
	}
	// gf2bv_pool_trim: returns the bytes given back to the device
	size_t trim(int device)
	{
		std::vector<void *> drop;
		size_t bytes = 0;
		{
			std::lock_guard<std::mutex> lk(mu);
			auto bf = big_free.find(device);
			if (bf != big_free.end()) for (const Big &b : bf->second) bytes += b.bytes;
			for (auto &kv : free_bufs) if (kv.first.first == device) bytes += kv.first.second * kv.second.size();
			take_idle_locked(device, drop);
		}
		for (void *q : drop) (void)hipFree(q);
		return bytes;
	}
	// hipMalloc; on out-of-memory everything idle on the device is freed and the allocation repeated once
	hipError_t malloc_retry(void **out, size_t bytes, int device)
	{
		hipError_t e = hipMalloc(out, bytes);
		if (e != hipErrorOutOfMemory) return e;
		(void)hipGetLastError();
		if (trim(device) == 0) return e;
		return hipMalloc(out, bytes);
	}
	hipError_t alloc(void **out, size_t bytes, int device)
	{
		bytes = std::max<size_t>(bytes, 16);
		if (bytes > kMaxBucket) {
			void *stale = nullptr;
			{
				std::lock_guard<std::mutex> lk(mu);
				auto &v = big_free[device];
				for (size_t i = 0; i < v.size(); i++)
					if (v[i].bytes >= bytes && v[i].bytes <= bytes + bytes / 2) {
						*out = v[i].p; big_live[v[i].p] = {device, v[i].bytes};
						v.erase(v.begin() + i);
						return hipSuccess;
					}
				if (v.size() >= kMaxBig) { stale = v.front().p; v.erase(v.begin()); }     // none fits: make room (the oldest goes) before allocating
			}
			if (stale) (void)hipFree(stale);
			hipError_t e = malloc_retry(out, bytes, device);
			if (e == hipSuccess && keep_big()) { std::lock_guard<std::mutex> lk(mu); big_live[*out] = {device, bytes}; }
			return e;
		}
		const size_t b = bucket(bytes);
		{
			std::lock_guard<std::mutex> lk(mu);
			auto &v = free_bufs[{device, b}];
			if (!v.empty()) {
				*out = v.back(); v.pop_back(); cached_bytes -= b;
				live[*out] = {device, b};
				return hipSuccess;
			}
		}
		hipError_t e = malloc_retry(out, b, device);
		if (e == hipSuccess) { std::lock_guard<std::mutex> lk(mu); live[*out] = {device, b}; }
		return e;
	}
	void release(void *p)
	{
		if (!p) return;
		std::vector<void *> drop;          // freed after the lock is gone: hipFree of a multi-GiB buffer synchronises the device
		{
			std::lock_guard<std::mutex> lk(mu);
			auto bl = big_live.find(p);
			auto it = live.find(p);
			if (bl != big_live.end()) {
				const int device = bl->second.first;
				const size_t bytes = bl->second.second;
				big_live.erase(bl);
				auto &v = big_free[device];
				Big nb; nb.p = p; nb.bytes = bytes;
				v.push_back(nb);
				size_t sum = 0;
				for (const Big &b : v) sum += b.bytes;
				const size_t cap = big_cap_locked(device);
				while (!v.empty() && (v.size() > kMaxBig || sum > cap)) { sum -= v.front().bytes; drop.push_back(v.front().p); v.erase(v.begin()); }     // the oldest go
			}
			else if (it != live.end()) {
				const auto key = it->second;
				live.erase(it);
				if (cached_bytes + key.second <= kMaxCached) { free_bufs[key].push_back(p); cached_bytes += key.second; }
				else drop.push_back(p);
			}
			else drop.push_back(p);
		}
		for (void *q : drop) (void)hipFree(q);
	}
	hipError_t event(hipEvent_t *e, bool timing)
	{
		int device = 0;
		(void)hipGetDevice(&device);
		{
			std::lock_guard<std::mutex> lk(mu);
			auto &v = events[{device, timing ? 1 : 0}];
			if (!v.empty()) { *e = v.back(); v.pop_back(); return hipSuccess; }
		}
		return timing ? hipEventCreate(e) : hipEventCreateWithFlags(e, hipEventDisableTiming);
	}
	void release_event(hipEvent_t e, bool timing)
	{
		if (!e) return;
		int device = 0;
		(void)hipGetDevice(&device);
		std::lock_guard<std::mutex> lk(mu);
		auto &v = events[{device, timing ? 1 : 0}];
		if (v.size() < 8192) v.push_back(e); else (void)hipEventDestroy(e);
	}
	// cls: 0 = normal priority, 1 = low priority (the bulk side of a solve); + 2 = the same for BATCH calls and their gangs.  The two
	// worlds do not trade streams (round 5): which low-priority stream a single solve finds on top of the pool decides whether it runs
	// 5.4 or 12.7 ms (low_stream_for below), and batch calls with several host threads return their streams in any order.
	hipError_t stream(hipStream_t *s, int device, int cls)
	{
		const bool low = cls & 1;
		{
			std::lock_guard<std::mutex> lk(mu);
			auto &v = streams[{device, cls}];
			if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; }
		}
		if (!low) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
		int lo = 0, hi = 0;
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
		return hipStreamCreateWithPriority(s, hipStreamNonBlocking, lo);
	}
	void release_stream(hipStream_t s, int device, int cls)
	{
		if (!s) return;
		std::lock_guard<std::mutex> lk(mu);
		streams[{device, cls}].push_back(s);
	}
	// A low-priority stream for the bulk side of a solve whose panel path runs on `a` (round 5).  Not every pair of streams is equal on
	// this chip: with some pairs the workgroups of a kernel on `a` are handed out at ~0.4 us apiece while a chip-filling kernel on the
	// other stream has its successor queued behind it -- 80 workgroups take 37 us instead of 3 -- and a latency-bound solve (MT19937
	// recovery, 78 blocks) runs 12.7 ms instead of 5.4: panel path and bulk path end up one after the other.  Which pairs: a property of
	// the two streams (the hardware queues behind them), stable for the life of the process, the same for every solve -- seen as "the
	// single solves after a batch call are slow this time": the batch had left another of its streams on top of the pool
	// (profiles/r05_stream_pairs.txt).  So a pair is probed once (two holds with the bulk update's footprint on the candidate, a
	// one-workgroup and an 80-workgroup kernel with a panel kernel's footprint on `a`: when did the last of the 80 begin) and the
	// verdict kept; idle streams known to be good with `a` are preferred, unknown ones probed, at most kMaxLow low-priority streams per
	// device are created, and when nothing passes the first candidate is taken as it is.  GF2BV_STREAM_PAIRS=0: any idle stream.
	static constexpr int kMaxLow = 10;
	std::mutex probe_mu;                                                  // one probe at a time (it wants the queues to itself)
	std::map<std::pair<hipStream_t, hipStream_t>, int> pair_ok;           // (panel stream, low-priority stream) -> 1 good, 0 throttled
	std::map<int, int> low_created;                                       // device -> low-priority streams created so far
	int pairs_probed = 0, pairs_bad = 0;
	// The probe: a miniature of a solve's pattern, NB "blocks".  Panel stream: two one-workgroup kernels with a panel kernel's footprint
	// (256 threads, 23 KiB of LDS) that work 10 us each, then a flag; bulk stream: a gate on that flag, then a chip-filling kernel with the
	// bulk update's footprint (512 threads and 133 KiB per CU) that holds 40 us.  The panel chain depends on nothing the bulk stream
	// does: on a good pair it ends after NB x ~24 us (measured 187-192 us for 8 blocks), on a bad one after NB x ~55 us (439-446): each
	// block's panel kernels wait for the hold before them.  (A first probe -- when does the last of 80 panel-shaped workgroups begin
	// beside two holds: 3 us or 36 -- saw only some of the bad pairs; this one agreed with the solve on 38 of 38 pairs.)
	hipError_t probe_pair(hipStream_t a, hipStream_t b, int device, int *ok, int depth = 0)
	{
		*ok = 1;
		constexpr int NB = 6;
		unsigned long long *q = nullptr, hq[2 * NB + 2];
		int *fl = nullptr;
		hipError_t e = alloc((void **)&q, sizeof hq, device);
		if (e != hipSuccess) return e;
		struct Free { Pool *p; void *q; ~Free() { p->release(q); } } g1{ this, q };
		if ((e = alloc((void **)&fl, sizeof(int) * 2 * NB, device)) != hipSuccess) return e;
		Free g2{ this, fl };
		if ((e = hipMemsetAsync(fl, 0, sizeof(int) * 2 * NB, a)) != hipSuccess) return e;
		if ((e = hipMemsetAsync(q, 0, sizeof hq, a)) != hipSuccess) return e;
		if ((e = hipStreamSynchronize(a)) != hipSuccess) return e;
		if ((e = hipStreamSynchronize(b)) != hipSuccess) return e;
		int cus = 256;
		(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
		k_probe_stamp_panel<<<dim3(1), dim3(256), 0, a>>>(q + 2 * NB);
		for (int i = 0; i < NB; i++) {
			k_probe_panel_work<<<dim3(1), dim3(256), 0, a>>>(q + 2 * i, 1000ull);
			k_probe_panel_work<<<dim3(1), dim3(256), 0, a>>>(q + 2 * i + 1, 1000ull);
			k_probe_set<<<dim3(1), dim3(1), 0, a>>>(fl + i);
			k_probe_wait<<<dim3(1), dim3(1), 0, b>>>(fl + i, fl + NB + i, 200000ull);
			k_probe_hold_bulk<<<dim3((unsigned)cus), dim3(512), 0, b>>>(q + 2 * NB + 1, 4000ull);
		}
		if ((e = hipGetLastError()) != hipSuccess) return e;
		if ((e = hipStreamSynchronize(a)) != hipSuccess) return e;
		if ((e = hipStreamSynchronize(b)) != hipSuccess) return e;
		if ((e = hipMemcpy(hq, q, sizeof hq, hipMemcpyDeviceToHost)) != hipSuccess) return e;
		const double chain_us = ((double)hq[2 * NB - 1] - (double)hq[2 * NB]) / 100.0;
		*ok = chain_us < 37.0 * NB;
		// a verdict near the threshold (good pairs measure ~24 us per probe block, bad ones ~55: other work on the device can push a
		// probe either way) is not kept on one sample: the pair is measured again, twice at most, and the FASTER run decides -- a
		// neighbour can only slow a probe down
		if (depth < 2 && chain_us > 0.75 * 37.0 * NB && chain_us < 1.5 * 37.0 * NB) {
			int again = 1;
			if (probe_pair(a, b, device, &again, depth + 1) == hipSuccess) *ok = *ok || again;
		}
		if (getenv("GF2BV_TRACE"))
			fprintf(stderr, "[gf2bv trace] stream pair %p / %p: the panel chain of %d probe blocks took %.1f us (%s)\n", (void *)a, (void *)b, NB, chain_us,
			        *ok ? "good" : "waits for the bulk stream");
		return hipSuccess;
	}
	// cls: the class the stream comes from and goes back to (1, or 3 for a batch call's gangs); also: the panel streams of the solves that
	// will run beside this one (all idle now) -- the stream has to get on with them as well.
	hipError_t low_stream_for(hipStream_t a, int device, hipStream_t *out, int cls = 1, const std::vector<hipStream_t> &also = {})
	{
		const bool pairing = !(getenv("GF2BV_STREAM_PAIRS") ? atoi(getenv("GF2BV_STREAM_PAIRS")) == 0 : plain_mode());
		if (!pairing || !a) return stream(out, device, cls);
		std::lock_guard<std::mutex> pl(probe_mu);
		std::vector<hipStream_t> rejected, with;
		with.push_back(a);
		for (hipStream_t x : also) if (x && x != a) with.push_back(x);
		hipError_t err = hipSuccess;
		*out = nullptr;
		// verdict of candidate c against everything in `with`, as far as known: 1 all good, 0 one throttled, -1 something unknown
		auto known = [&](hipStream_t c) {
			int r = 1;
			for (hipStream_t x : with) {
				auto it = pair_ok.find({x, c});
				if (it == pair_ok.end()) r = r == 0 ? 0 : -1; else if (it->second == 0) r = 0;
			}
			return r;
		};
		for (int attempt = 0; attempt < kMaxLow + 2 && !*out; attempt++) {
			hipStream_t c = nullptr;
			bool known_good = false;
			{
				std::lock_guard<std::mutex> lk(mu);
				auto &v = streams[{device, cls}];
				for (size_t i = v.size(); i-- > 0 && !c;)
					if (known(v[i]) == 1) { c = v[i]; v.erase(v.begin() + (ptrdiff_t)i); known_good = true; }
				for (size_t i = v.size(); i-- > 0 && !c;)
					if (known(v[i]) == -1) { c = v[i]; v.erase(v.begin() + (ptrdiff_t)i); }
				if (!c) {
					if (low_created[device] >= kMaxLow) break;
					low_created[device]++;
				}
			}
			if (known_good) { *out = c; break; }
			if (!c) {
				int lo = 0, hi = 0;
				(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
				if ((err = hipStreamCreateWithPriority(&c, hipStreamNonBlocking, lo)) != hipSuccess) break;
			}
			int all_ok = 1;
			for (hipStream_t x : with) {
				int ok = -1;
				{
					std::lock_guard<std::mutex> lk(mu);
					auto it = pair_ok.find({x, c});
					if (it != pair_ok.end()) ok = it->second;
				}
				if (ok < 0) {
					ok = 1;
					if (probe_pair(x, c, device, &ok) != hipSuccess) { (void)hipGetLastError(); ok = 1; }      // (a probe that cannot run decides nothing -- and leaves no verdict: the pair is asked again)
					else {
						std::lock_guard<std::mutex> lk(mu);
						pair_ok[{x, c}] = ok;
						pairs_probed++; pairs_bad += !ok;
					}
				}
				if (!ok) { all_ok = 0; break; }
			}
			if (all_ok) *out = c; else rejected.push_back(c);
		}
		if (!*out && !rejected.empty()) { *out = rejected.front(); rejected.erase(rejected.begin()); }
		for (hipStream_t r : rejected) release_stream(r, device, cls);
		if (!*out && err == hipSuccess) return stream(out, device, cls);
		return err;
	}
};
Pool &pool()
{
	static Pool *p = new Pool();      // intentionally leaked: the HIP runtime may be gone at static destruction
	return *p;
}

// ---- kernel configurations -----------------------------------------------------------------
// Bulk update: G panels fused per HBM pass, T grease tables per panel (balanced bit-fields).
int streams_run_concurrently(int device, hipStream_t a, hipStream_t b, int *ok);
int xcd_dispatch_is_round_robin(int device, hipStream_t st, int *ok);
void forget_concurrency(int device);

struct UpdateImpl {
	int G, T, lds_bytes, threads;
	hipError_t (*update)(dim3, hipStream_t, u64 *, i64, i64, int, int, int, const PanelRec *, const PanelAux *,
	                     const u64 *, const int *, int, int, int, int, int, bool, const uint4 *, SysStride, hipEvent_t, hipEvent_t, int);
};

// 16-byte tiles: G = 4 panels, T = 8 byte fields per panel.  nw_lo < 0 selects the HALF instance (only the tile's second
// word is stored: its first one belongs to the next block's window); the window's whole tiles are simply not launched.
template <int NT, int DEPTH, bool PIPE, int LB>
hipError_t launch_update16(dim3 grid, hipStream_t s, u64 *M, i64 rows, i64 srows, int j0, int gb, int wlo,
                           const PanelRec *panels, const PanelAux *aux, const u64 *multset, const int *blk_first,
                           int tile_begin, int ntiles, int world, int wrank, int nw_lo, bool stream_rows, const uint4 *Pc, SysStride ss,
                           hipEvent_t begun, hipEvent_t done, int xcd_nsys)
{
	if (xcd_nsys > 0) grid = dim3(grid.x * grid.y);          // one line of workgroups, decoded in the kernel (a system per XCD)
	{
		// stream_rows: streaming (non-temporal) row accesses -- pinned gangs (GF2BV_GANG_NT=0: plain).  The streaming form exists for the
		// default instance only: with GF2BV_UPDATE != 0 the selected instance runs with plain accesses.
		if (stream_rows && NT == 512 && DEPTH == 3 && PIPE && LB == 512) {
			if (nw_lo < 0)
				hipExtLaunchKernelGGL((k_update16<512, true, 3, true, 512, true>), grid, dim3(512), 0, s, begun, done, 0, M, rows, srows, j0, gb, wlo,
				                      panels, aux, multset, blk_first, tile_begin, ntiles, world, wrank, Pc, ss, xcd_nsys);
			else
				hipExtLaunchKernelGGL((k_update16<512, false, 3, true, 512, true>), grid, dim3(512), 0, s, begun, done, 0, M, rows, srows, j0, gb, wlo,
				                      panels, aux, multset, blk_first, tile_begin, ntiles, world, wrank, Pc, ss, xcd_nsys);
			return hipGetLastError();
		}
	}
	if (nw_lo < 0)
		hipExtLaunchKernelGGL((k_update16<NT, true, DEPTH, PIPE, LB>), grid, dim3(NT), 0, s, begun, done, 0, M, rows, srows, j0, gb, wlo,
		                      panels, aux, multset, blk_first, tile_begin, ntiles, world, wrank, Pc, ss, xcd_nsys);
	else
		hipExtLaunchKernelGGL((k_update16<NT, false, DEPTH, PIPE, LB>), grid, dim3(NT), 0, s, begun, done, 0, M, rows, srows, j0, gb, wlo,
		                      panels, aux, multset, blk_first, tile_begin, ntiles, world, wrank, Pc, ss, xcd_nsys);
	return hipGetLastError();
}
#define UPDATE16_IMPL(NT, D, P, LB) { 4, 8, 2 * 256 * 256 + GF2_GMAX * 64 * 20, NT, launch_update16<NT, D, P, LB> }

const UpdateImpl kUpdates[] = {
	// default: 8 wavefronts (two per SIMD, 157 VGPRs: three batches of 64 rows in flight per wavefront, the lookups of
	// round r+1 issued before the XORs of round r); the kernel is bound by the memory side, so more wavefronts buy
	// nothing (65536^2 / 131072^2 wall: 37.8 / 212.9 ms; 12 wavefronts 38.4 / 212.5; 16: 40.7 / 221) and half of every
	// SIMD's registers plus 27 KiB of LDS stay free for the panel kernels of the next block
	UPDATE16_IMPL(512, 3, true, 512),
	UPDATE16_IMPL(768, 2, true, 768),
	UPDATE16_IMPL(768, 2, true, 1024),
	UPDATE16_IMPL(1024, 2, false, 1024),
	UPDATE16_IMPL(768, 3, true, 768),
	UPDATE16_IMPL(640, 2, true, 640),
};

const UpdateImpl *pick_update()
{
	const UpdateImpl *chosen = &kUpdates[0];
	if (const char *e = getenv("GF2BV_UPDATE")) {      // index into the table above (tests run every instance)
		const int idx = atoi(e);
		if (idx >= 0 && idx < (int)(sizeof kUpdates / sizeof kUpdates[0])) chosen = &kUpdates[idx];
	}
	return chosen;
}

// Back-substitution sweeps over Y use the single-panel kernel (5-bit tables, 128-byte segments).
constexpr int YK = 5, YTW = 16, YNT = 1024;
hipError_t launch_ysweep(dim3 grid, hipStream_t s, u64 *Y, i64 ys, i64 rows, const PanelRec *rec,
                         const u64 *mult, int ntiles, int rpb)
{
	constexpr int lds = SweepCfg<YK, YTW>::LDS_BYTES;
	static std::mutex attr_mu;
	static std::map<int, bool> attr_set;          // device -> the > 64 KiB dynamic-LDS attribute has been raised there
	int dev = 0;
	(void)hipGetDevice(&dev);
	{
		std::lock_guard<std::mutex> lk(attr_mu);
		if (!attr_set[dev]) {
			hipError_t e = hipFuncSetAttribute((const void *)k_sweep<YK, YTW, YNT>,
			                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds);
			if (e != hipSuccess) return e;
			attr_set[dev] = true;
		}
	}
	k_sweep<YK, YTW, YNT><<<grid, dim3(YNT), lds, s>>>(Y, ys, rows, rec, 1, mult, 0, ntiles, rpb);
	return hipGetLastError();
}

constexpr int TW = GF2_TW;        // words per column tile

// ---- one solve ---------------------------------------------------------------------------------
struct Solver {
	int device = 0;
	hipStream_t sA = nullptr, sB = nullptr;      // panel path / bulk path
	bool own_sA = false, own_sB = false;
	u64 *M = nullptr;             // tile-major working copy (always owned)
	const u64 *src = nullptr;     // caller's row-major matrix on the device (stride words per row)
	u64 *tmp_src = nullptr;       // row-major staging buffer when the input came from the host
	u64 *Ybuf = nullptr;
	u64 *Minv = nullptr;          // inverted diagonal blocks of the back-substitution (k_bs_inv -> k_bs_near2)
	i64 rows = 0, cols = 0, stride = 0;
	i64 ntiles = 0, srows = 0;    // tiles, rows per tile slab (padded)
	// gang: nsys same-shape systems eliminated in lock-step by the same launches (blockIdx.y = system);
	// system s lives at M + s * m_stride words / arena + s * arena_stride bytes, its input at src + s * src_sys_words
	int nsys = 1;
	int gang_nsys = 1;            // (a view: the size of the gang it belongs to)
	// column-slab solve of ONE system over `world` GPUs: this rank owns the column tiles t with t % world == wrank
	// (cyclic, so that the shrinking trailing matrix stays balanced) and runs the bulk path on those only; the panel
	// path of a block runs on the owner of its window's tile (see gf2bv_slab_* below)
	int world = 1, wrank = 0;
	bool ext_M = false;           // the working matrix belongs to the caller (slab solves: a tensor the ranks exchange tiles of)
	i64 m_stride = 0, src_sys_words = 0;
	size_t arena_stride = 0;
	bool view = false;            // a non-owning window on one system of a gang (back-substitution, export)
	bool own_bs = true;           // (a view) Y / ycols / out are its own; false: `out` points into the gang's (enqueue_backward_gang)
	SysStride ss() const { return SysStride{ m_stride, (i64)arena_stride }; }
	int mode = 0;
	bool time_kernels = false;
	int dbg_sync = 0;
	const UpdateImpl *impl = nullptr;

	void *arena = nullptr;        // st, panels, aux, fu, alive, pivcol, urow, blk_first, mult, Wb live in here
	SolveState *st = nullptr;
	PanelRec *panels = nullptr;
	PanelAux *aux = nullptr;
	FindUnit *fu = nullptr;
	int *died = nullptr;          // per row: panel that made it a pivot source, GF2_NEVER while alive
	int *pivcol = nullptr, *urow = nullptr, *blk_first = nullptr;
	u64 *mult = nullptr;          // nsets sets x G x rows: block b writes / reads set b % nsets (2 = ping-pong; an outer panel of the
	                              // two-level elimination keeps all its K blocks' multipliers until its outer pass: nsets = K)
	int nsets = 2;
	// two-level elimination (large systems; see k_update16k): blocks [0, tl_bend) go in outer panels of tl_K blocks
	int tl_K = 0, tl_bend = 0;
	i64 tile_hi = 0;              // bulk kernels touch tiles < tile_hi (= ntiles; the outer panel's end while it is eliminated)
	static constexpr int nlist = 2;      // row lists / T matrices kept: by panel parity (the previous panel's outer pass may still be reading its own)
	hipStream_t sC = nullptr;     // outer passes: panel p's runs BESIDE the inner elimination of panel p + 1 (sA + sB)
	bool outer_side = true;       // GF2BV_OUTER_SIDE=0: the outer step on the next panel's tiles in front of the pass on the outer stream (rounds 3-5)
	                              // (late round 5 also split the pass itself over two streams -- GF2BV_OUTER_SPLIT: 131072^2 162 -> 155 ms, 262144^2
	                              // -0.8 %, but a fourth stream that cost later batch calls of the process their overlap; removed in round 6,
	                              // profiles/r05_outer_shapes.txt)
	hipEvent_t evOuter = nullptr, evPri = nullptr, evPanelDone = nullptr, evBig = nullptr;
	bool bulk_waits_outer = false;     // the next bulk launch of the one-level schedule has to wait for the last outer pass
	bool ends_outer_panel(int b) const { return tl_K > 0 && b < tl_bend && (b + 1) % tl_K == 0; }
	u64 *Wb = nullptr;            // 2 x rows x GMAX window words (the panel steps ping-pong between the halves)
	u64 *Uwin = nullptr;          // rank x GMAX: pivot rows' words of the following window (k_prio_window -> k_unwind)
	int *oprow = nullptr;         // k_outer_prow -> k_outer_apply / k_update16k: row lists of the outer panel being applied
	u64 *Tm = nullptr;            // k_outer_trsm<IDENT> -> k_outer_apply: the panel's pivot rows as combinations of its source rows
	bool outer_chain = false;     // GF2BV_OUTER_CHAIN=1: the chain itself on every word group instead (the first form; tests)
	u64 *Pc = nullptr;            // final pivot rows of the current block, compact: [tile][panel][pivot bit] x 16 B (k_block_trsm -> k_update16)
	u64 *Pfast = nullptr;         // scratch of k_block_fast: the pivot rows' window words of a block, [panel][word][column]
	SyncFlags *sf = nullptr;      // progress counters of the two streams (k_gate)
	bool flag_sync = true;        // per-block hand-overs between the streams through sf + k_gate instead of events (GF2BV_FLAG_SYNC=0)
	int sync_base = 0;            // the counters only grow: a pass that re-enqueues blocks (resume after a poisoned one) counts from here
	bool fast_blocks = true;      // try the one-launch block search on dense blocks (GF2BV_FAST=0 disables)
	bool optimistic = true;       // ... and drop the general panel steps behind it once block 0 has taken it (GF2BV_OPTIMISTIC=0)
	int units = 0;
	bool ext_events = true;       // hand-off and timing events ride on kernel start / completion signals (hipExtLaunchKernel)
	                              // instead of marker packets: ~1 % at every size; GF2BV_EXT_EVENTS=0 restores hipEventRecord
	int sparse_mode = 2;          // search skips absent columns: 0 never, 1 always, 2 per chunk by density (GF2BV_SPARSE)
	int fused_rpt = 0;            // row blocks of 256 per narrowing workgroup of k_block_fast_narrow (0 = by size)
	bool use_pc = true;           // GF2BV_PC=0: the bulk update fetches the pivot rows through the panel records, as before round 3
	bool xcd_pin = true;          // gangs of a multiple of 8 systems: every system's bulk-update workgroups on ONE XCD, one system after the
	                              // other there, xcd_wgs workgroups each (GF2BV_XCD_WGS); GF2BV_XCD_PIN=0: the plain (spans, systems) grid
	int xcd_wgs = 32;
	bool nt_gang = true;          // streaming (non-temporal) row accesses of the bulk update of pinned gangs (GF2BV_GANG_NT=0: plain); single systems
	                              // keep plain accesses (measured <= 1 % either way in round 4)
	bool gang_bs = true;          // GF2BV_GANG_BS=0: one back-substitution chain per system of a gang, as rounds 1-3
	// sparse systems (round 5): blocks the dense one-launch search cannot take go through k_block_sparse -- candidates = the alive rows
	// with a non-zero window, from the bit masks the look-ahead leaves in wmask (GF2BV_SPARSE_FAST=0: the general panel steps)
	bool sparse_fast = true;      // allowed at all
	bool sparse_on = false;       // this solve has switched to it (after block 0 went the general way)
	int sparse_giveups = 0;
	int sparse_tier = 0;          // pool size of k_block_sparse: 0 = 1024 rows (beside the bulk update), 1 = 4096, 2 = 6144
	u64 *wmask = nullptr;         // 2 x ceil(rows / 64) words: alive / alive with a non-zero window, per 64 rows
	bool fused_narrow = true;     // optimistic blocks: search and narrow step in ONE launch (k_block_fast_narrow); GF2BV_FUSED_NARROW=0: two
	int narrow_rpt = 1;           // row blocks of 256 per narrow workgroup of a panel step (set in solver_alloc; GF2BV_NARROW_RPT)
	int self_wait = 5000;         // ticks (100 MHz) unit 0 of a panel search waits for the other units before it leaves
	                              // publishing to the last arriver: 50 us (GF2BV_SELF_WAIT_US; 0 = never wait)
	u64 *Y = nullptr;
	int *ycols = nullptr;
	u64 *out = nullptr;
	i64 ys = 0;
	int ny = 0;
	int h_ycol = 0;               // (host copy of the gang's one right-hand-side column: lives as long as the copy may)
	i64 maxr = 0;
	int npanels = 0, nblocks = 0;
	i64 wt = 0, cw = 0;

	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
	std::vector<hipEvent_t> evA, evPrio, kev, waitPrio;     // waitPrio[b]: the event that means "bulk of block b complete"
	std::vector<int> free_order;     // free columns in M4RI kernel order (mode 1)
	// host staging of the export (filled by asynchronous copies between finish_begin and finish_end)
	SolveState hst{};
	std::vector<u64> hout;
	std::vector<PanelRec> hp;
	std::vector<int32_t> hpiv;
	hipEvent_t evx = nullptr;
	std::chrono::steady_clock::time_point t_begin;
	float ms_pack = 0;

	~Solver() { release(); }
	void release()
	{
		if (!arena && !M && !tmp_src && !Y && !ycols && !out && !sA && !sB && !ev0) return;     // nothing held
		(void)hipSetDevice(device);
		if (view) {               // owns only what its own back-substitution allocated
			if (sA && (Y || ycols || out)) (void)hipStreamSynchronize(sA);
			Pool &P = pool();
			if (own_bs) for (void *p : { (void *)Y, (void *)ycols, (void *)out, (void *)Minv }) P.release(p);
			P.release_event(ev2, true);
			P.release_event(evx, true); evx = nullptr;
			Y = nullptr; ycols = nullptr; out = nullptr; Minv = nullptr; ev2 = nullptr;
			arena = nullptr; M = nullptr; tmp_src = nullptr; sA = sB = nullptr;
			ev0 = ev1 = ev3 = nullptr;
			kev.clear(); evA.clear(); evPrio.clear();
			return;
		}
		// nothing goes back to the pool while work may still be in flight (error paths return early)
		if (sC) (void)hipStreamSynchronize(sC);
		if (sB) (void)hipStreamSynchronize(sB);
		if (arena || M) (void)hipStreamSynchronize(sA);
		Pool &P = pool();
		for (void *p : { arena, (void *)Y, (void *)ycols, (void *)out, (void *)Minv, (void *)(ext_M ? nullptr : M), (void *)tmp_src }) P.release(p);
		arena = nullptr; Y = nullptr; ycols = nullptr; out = nullptr; Minv = nullptr; M = nullptr; tmp_src = nullptr;
		st = nullptr; panels = nullptr; aux = nullptr; fu = nullptr; died = nullptr; pivcol = nullptr;
		urow = nullptr; blk_first = nullptr; mult = nullptr; Wb = nullptr;
		for (hipEvent_t *e : { &ev0, &ev1, &ev2, &ev3, &evx }) { P.release_event(*e, true); *e = nullptr; }
		for (hipEvent_t *e : { &evOuter, &evPri, &evPanelDone, &evBig }) { P.release_event(*e, false); *e = nullptr; }
		if (sC) P.release_stream(sC, device, nsys > 1 ? 3 : 1);
		sC = nullptr;
		for (hipEvent_t e : kev) P.release_event(e, true);
		for (hipEvent_t e : evA) P.release_event(e, false);
		for (hipEvent_t e : evPrio) P.release_event(e, false);
		kev.clear(); evA.clear(); evPrio.clear();
		if (own_sB && sB) P.release_stream(sB, device, nsys > 1 ? 3 : 1);
		sB = nullptr;
		if (own_sA && sA) P.release_stream(sA, device, false);
		sA = nullptr;
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

// Pool resources of an entry point that must go back on every return path (error paths included).
struct Scratch {
	std::vector<void *> bufs;
	std::vector<hipEvent_t> timing_events;
	hipStream_t sync_first = nullptr;          // synchronised before anything is released (work may still be in flight)
	~Scratch()
	{
		if (sync_first) (void)hipStreamSynchronize(sync_first);
		for (void *p : bufs) pool().release(p);
		for (hipEvent_t e : timing_events) pool().release_event(e, true);
	}
	hipError_t alloc(void **out, size_t bytes, int device)
	{
		hipError_t e = pool().alloc(out, bytes, device);
		if (e == hipSuccess) bufs.push_back(*out);
		return e;
	}
	hipError_t event(hipEvent_t *ev)
	{
		hipError_t e = pool().event(ev, true);
		if (e == hipSuccess) timing_events.push_back(*ev);
		return e;
	}
};

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

// Two-level elimination: which blocks go in outer panels, and of how many blocks.
// An outer panel of K blocks saves HBM round trips -- the rows right of it make one trip per K blocks, and the outer pass runs
// 4.5-4.9 TB/s of sweep-words inside a solve (K = 8) where k_update16 runs 3.6-4.0 -- and costs ~0.4 ms of small launches per
// panel (k_outer_apply on two tile ranges, the priority part of the pass) plus a slower inner elimination beside the pass.
// Measured (K = 8 from the first block to the last full panel): 262144^2 1.517 -> 1.277 s, 131072^2 203 -> 195 ms, 98304^2 89 ->
// 100 ms, 65536^2 34 -> 47 ms; with a threshold on what remains right of the panel (profiles/r03_two_level.txt): 131072^2 best at
// 384-512 MiB (182 ms), 98304^2 at 512 MiB (87.3 against 92.8), 65536^2 never.  So: K = 8 while more than 512 MiB remain
// (262144^2: the first 75 % of the pivots; 131072^2: the first half; 98304^2: the first third; nothing below ~70000^2).
// GF2BV_TWO_LEVEL=0 turns it off, =K (2..12)
// forces K from the first block on for every full panel whatever the size (tests).  Single systems on one GPU only: gangs
// and column-slab solves keep the one-level schedule.
void plan_two_level(Solver &S)
{
	S.tl_K = 0; S.tl_bend = 0; S.nsets = 2;
	// (a column-slab handle -- ext_M, at world size 1 too -- is driven block by block through slab_factor_on / slab_apply_on, which
	// know nothing of outer panels: with a plan it would skip the look-ahead at every panel end and never run an outer pass)
	if (S.world != 1 || S.ext_M || S.impl->G != GF2_GMAX) return;
	// gangs keep the one-level schedule.  Round 4 built outer panels for gangs (every outer kernel takes blockIdx.y = system) as an
	// opt-in, bit-exact and slower: 4.4 against 3.9 ms per system; round 5 measured it again on the final gang layout (a system per
	// XCD, streaming row accesses, gangs of 32 and of 8): 3.50-3.60 against 3.20-3.29 ms per system of 32768^2, K = 4 / 8 / 12 and
	// thresholds alike (profiles/r05_batch_scans.txt) -- a gang's inner elimination is a chain of all-rows launches of thousands of
	// workgroups that do not fit beside k_update16k, so chain and pass wait for each other instead of overlapping.  The knob
	// (GF2BV_GANG_TWO_LEVEL) and its test are gone with round 5.
	if (S.nsys != 1) return;
	// outer panels of 12 blocks from 3 GiB up, of 8 below: the outer pass gains with K (isolated 5.20 / 5.30 / 5.38 TB/s of
	// sweep-words for K = 8 / 10 / 12), the inner elimination and the T chain grow with it -- 262144^2 1.303 -> 1.269 s,
	// 196608^2 566 -> 557 ms, 393216^2 4.26 -> 4.13 s, but 131072^2 184.0 -> 185.3 ms (profiles/r03_two_level.txt)
	int K = (double)S.rows * (double)S.wt * 8.0 >= 3.0 * 1073741824.0 ? GF2_KMAX : 8;       // (per system: a gang's panels stay at 8 blocks)
	double min_bytes = 0.5 * 1073741824.0;
	if (const char *e = getenv("GF2BV_TWO_LEVEL"); e && *e) {
		const int v = atoi(e);
		if (v <= 0) return;
		K = std::min(GF2_KMAX, std::max(2, v));
		min_bytes = 0;
	}
	if (const char *e = getenv("GF2BV_TWO_LEVEL_MIN_MIB"); e && *e) min_bytes = 1048576.0 * atof(e);      // (threshold scans)
	const int G = S.impl->G;
	int bend = 0;
	for (int b0 = 0; b0 + K < S.nblocks; b0 += K) {            // (the last block never ends an outer panel: it may be short)
		const i64 rows_left = S.rows - (i64)(b0 + K) * 64 * G, words_left = S.wt - (i64)(b0 + K) * G;
		if ((i64)(b0 + K) * 64 * G > S.cols || rows_left <= 0 || words_left <= 0) break;
		if ((double)rows_left * (double)words_left * 8.0 * (double)S.nsys < min_bytes) break;
		bend = b0 + K;
	}
	if (!bend) return;
	S.tl_K = K; S.tl_bend = bend;
	if (const char *e = getenv("GF2BV_OUTER_CHAIN"); e && *e) S.outer_chain = atoi(e) != 0;
	S.nsets = 2 * K;              // the outer pass of panel p reads its K sets while the blocks of panel p + 1 write theirs
}

int solver_alloc(Solver &S)
{
	S.wt = (S.cols + 1 + 63) / 64;
	S.cw = (S.cols + 63) / 64;
	S.npanels = (int)((S.cols + 63) / 64);
	S.maxr = std::min(S.rows, S.cols);
	S.impl = pick_update();
	S.ntiles = tiles_for(S.wt);
	S.srows = slab_rows(S.rows);
	S.m_stride = S.ntiles * TW * S.srows;
	if (!S.M) HIPCHK(pool().alloc((void **)&S.M, sizeof(u64) * S.m_stride * S.nsys + kOuterSlackBytes, S.device));
	if (S.src && S.rows > 0) {
		k_to_tiled<<<dim3((unsigned)((S.rows + 63) / 64), (unsigned)((S.ntiles + 15) / 16), S.nsys), dim3(256), 0, S.sA>>>(
			S.src, S.stride, S.rows, S.ntiles, std::min(S.wt, S.stride), S.srows, S.M, S.src_sys_words, S.ss());
	}
	const int G = S.impl->G;
	S.nblocks = (S.npanels + G - 1) / G;
	S.tile_hi = S.ntiles;
	plan_two_level(S);
	S.units = (int)std::min<i64>(256, std::max<i64>(1, (S.rows + 255) / 256));
	if (const char *e = getenv("GF2BV_DEBUG_SYNC")) S.dbg_sync = atoi(e);
	if (const char *e = getenv("GF2BV_FAST")) S.fast_blocks = atoi(e) != 0;
	const bool plain = plain_mode();
	S.flag_sync = S.world == 1 && !plain; // (a column-slab solve hands over through the host between the pieces: events)
	S.optimistic = !plain;
	if (const char *e = getenv("GF2BV_FLAG_SYNC")) S.flag_sync = S.flag_sync && atoi(e) != 0;
	if (getenv("GF2BV_SERIAL")) S.flag_sync = false;      // (one stream: the panel gate would wait for a gate queued behind it)
	if (const char *e = getenv("GF2BV_OPTIMISTIC")) S.optimistic = atoi(e) != 0;
	if (const char *e = getenv("GF2BV_FUSED_NARROW")) S.fused_narrow = atoi(e) != 0;
	if (const char *e = getenv("GF2BV_SPARSE_FAST")) S.sparse_fast = atoi(e) != 0;
	if (const char *e = getenv("GF2BV_PC")) S.use_pc = atoi(e) != 0;
	if (const char *e = getenv("GF2BV_XCD_PIN")) S.xcd_pin = atoi(e) != 0;
	else if (plain) S.xcd_pin = false;
	else if (S.nsys >= 8 && S.nsys % 8 == 0) {           // a gang that would be pinned: is the dispatch order what the pinning assumes?
		int ok = 0;
		int rc = xcd_dispatch_is_round_robin(S.device, S.sA, &ok);
		if (rc) return rc;
		S.xcd_pin = ok != 0;
	}
	if (const char *e = getenv("GF2BV_XCD_WGS")) S.xcd_wgs = std::min(256, std::max(1, atoi(e)));
	if (const char *e = getenv("GF2BV_GANG_NT")) S.nt_gang = atoi(e) != 0;
	if (const char *e = getenv("GF2BV_GANG_BS")) S.gang_bs = atoi(e) != 0;
	// narrow workgroups: as many rows each as keeps ~256 of them (all systems of a gang together) busy, at most 8 blocks
	{
		const i64 blocks = (S.rows + 255) / 256 * std::max(1, S.nsys);
		S.narrow_rpt = (int)std::min<i64>(8, std::max<i64>(1, blocks / 256));
	}
	if (const char *e = getenv("GF2BV_SELF_WAIT_US")) { int v = atoi(e); if (v >= 0 && v <= 1000000) S.self_wait = v * 100; }
	if (getenv("GF2BV_SERIAL")) { S.sB = S.sA; S.own_sB = false; }     // ablation: no look-ahead overlap
	else {
		// the bulk path yields to the (latency-critical) panel path wherever both have work queued: lowest priority
		// (round 5: single systems take a stream that is KNOWN to run beside sA, see Pool::low_stream_for.  Gangs take any: two
		// lock-step gangs side by side ran 16 MT19937 systems in 33 ms on whatever the pool handed out and in 36-41 ms on pairs
		// chosen this way -- with four busy queues other relations than panel / bulk of ONE solve decide, profiles/r05_stream_pairs.txt)
		if (S.nsys == 1) HIPCHK(pool().low_stream_for(S.sA, S.device, &S.sB));
		else HIPCHK(pool().stream(&S.sB, S.device, 3));
		S.own_sB = true;
	}
	if (S.flag_sync && S.sB != S.sA) {
		int ok = 0;
		int rc = streams_run_concurrently(S.device, S.sA, S.sB, &ok);
		if (rc) return rc;
		S.flag_sync = ok != 0;
	}
	// one arena for all side arrays (a dozen hipMalloc/hipFree pairs cost more than a small solve)
	{
		const i64 R = std::max<i64>(1, S.rows), NP = std::max(1, S.npanels);
		size_t off = 0;
		auto carve = [&](size_t bytes) { size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
		const size_t o_st = carve(sizeof(SolveState)), o_sf = carve(sizeof(SyncFlags)), o_pan = carve(sizeof(PanelRec) * NP), o_aux = carve(sizeof(PanelAux) * NP),
		             o_fu = carve(sizeof(FindUnit) * (S.units + 1 + GF2_MAXGROUPS)), o_alive = carve(sizeof(int) * (size_t)R),
		             o_piv = carve(sizeof(int) * (S.maxr + 64)), o_urow = carve(sizeof(int) * (S.maxr + 64)),
		             o_blk = carve(sizeof(int) * std::max(1, S.nblocks)), o_mult = carve(sizeof(u64) * S.nsets * G * mult_rows(R) + (S.tl_K ? kOuterSlackBytes : 0)),
		             o_wb = carve(sizeof(u64) * 2 * GF2_GMAX * R), o_uw = carve(sizeof(u64) * GF2_GMAX * (S.maxr + 64)),
		             o_pf = carve(sizeof(u64) * GF2_GMAX * GF2_GMAX * 64), o_opr = carve(sizeof(int) * GF2_OUTER_LISTS * S.nlist),
		             o_tm = carve(S.tl_K ? S.nlist * sizeof(u64) * GF2_KMAX * GF2_KMAX * GF2_GMAX * 64 * GF2_GMAX : 0),
		             o_pc = carve(S.use_pc ? sizeof(u64) * 2 * GF2_GMAX * 64 * (size_t)S.ntiles : 0),
		             o_wm = carve(sizeof(u64) * 2 * (size_t)((R + 63) / 64 + 1));
		S.arena_stride = off;
		S.sync_base = 0;
		HIPCHK(pool().alloc(&S.arena, off * S.nsys, S.device));
		char *base = (char *)S.arena;
		S.st = (SolveState *)(base + o_st); S.sf = (SyncFlags *)(base + o_sf); S.panels = (PanelRec *)(base + o_pan); S.aux = (PanelAux *)(base + o_aux);
		S.fu = (FindUnit *)(base + o_fu); S.died = (int *)(base + o_alive); S.pivcol = (int *)(base + o_piv);
		S.urow = (int *)(base + o_urow); S.blk_first = (int *)(base + o_blk); S.mult = (u64 *)(base + o_mult);
		S.Wb = (u64 *)(base + o_wb); S.Uwin = (u64 *)(base + o_uw); S.Pfast = (u64 *)(base + o_pf); S.oprow = (int *)(base + o_opr); S.Tm = (u64 *)(base + o_tm);
		S.Pc = S.use_pc ? (u64 *)(base + o_pc) : nullptr;
		S.wmask = (u64 *)(base + o_wm);
		// zero everything that is read before it is written: state, panel records, unit scratch, block bounds, multipliers
		for (int s = 0; s < S.nsys; s++) {
			char *b = base + (size_t)s * off;
			HIPCHK(hipMemsetAsync(b + o_st, 0, o_alive - o_st, S.sA));
			HIPCHK(hipMemsetAsync(b + o_alive, GF2_NEVER & 0xff, sizeof(int) * (size_t)R, S.sA));
			HIPCHK(hipMemsetAsync(b + o_blk, 0, o_wb - o_blk, S.sA));
		}
	}
	HIPCHK(pool().event(&S.ev0, true));
	HIPCHK(pool().event(&S.ev1, true));
	HIPCHK(pool().event(&S.ev2, true));
	HIPCHK(pool().event(&S.ev3, true));
	if (S.tl_K) {
		for (hipEvent_t *e : { &S.evOuter, &S.evPri, &S.evPanelDone }) HIPCHK(pool().event(e, false));
		if (S.sB != S.sA) { if (S.nsys == 1) HIPCHK(pool().low_stream_for(S.sA, S.device, &S.sC)); else HIPCHK(pool().stream(&S.sC, S.device, 3)); }       // (GF2BV_SERIAL: everything on one stream)
		HIPCHK(pool().event(&S.evBig, false));
		if (const char *e = getenv("GF2BV_OUTER_SIDE"); e && *e) S.outer_side = atoi(e) != 0;
	}
	S.evA.resize(S.nblocks); S.evPrio.resize(S.nblocks); S.waitPrio.assign(S.nblocks, nullptr);
	for (int b = 0; b < S.nblocks; b++) {
		HIPCHK(pool().event(&S.evA[b], false));
		HIPCHK(pool().event(&S.evPrio[b], false));
	}
	return GF2BV_OK;
}

// Do kernels of two streams execute at the same time on this device, in this process?  (See k_probe_wait.)  Asked once per
// device; GF2BV_FLAG_SYNC=0 skips the question and the flag hand-over with it.
struct ConcurrencyCache { std::mutex mu; std::map<int, int> known; };
ConcurrencyCache &concurrency_cache() { static ConcurrencyCache c; return c; }
void forget_concurrency(int device)       // a gate timed out although the probe had passed: events from now on
{
	ConcurrencyCache &c = concurrency_cache();
	std::lock_guard<std::mutex> lk(c.mu);
	c.known[device] = 0;
}
int streams_run_concurrently(int device, hipStream_t a, hipStream_t b, int *ok)
{
	std::mutex &mu = concurrency_cache().mu;
	std::map<int, int> &known = concurrency_cache().known;
	std::lock_guard<std::mutex> lk(mu);
	auto it = known.find(device);
	if (it != known.end()) { *ok = it->second; return GF2BV_OK; }
	// (GF2BV_FLAG_SYNC=2: take concurrency for granted until a gate times out -- exercises the time-out / retry path)
	if (getenv("GF2BV_FLAG_SYNC") && atoi(getenv("GF2BV_FLAG_SYNC")) == 2) { *ok = 1; return GF2BV_OK; }
	int *d = nullptr;
	HIPCHK(pool().alloc((void **)&d, 2 * sizeof(int), device));
	struct Free { int *p; ~Free() { pool().release(p); } } guard{d};
	HIPCHK(hipMemsetAsync(d, 0, 2 * sizeof(int), a));
	HIPCHK(hipStreamSynchronize(a));
	HIPCHK(hipStreamSynchronize(b));
	k_probe_wait<<<dim3(1), dim3(1), 0, a>>>(d, d + 1, 200000ull);      // 2 ms
	k_probe_set<<<dim3(1), dim3(1), 0, b>>>(d);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(a));
	HIPCHK(hipStreamSynchronize(b));
	int h[2] = { 0, 0 };
	HIPCHK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
	*ok = known[device] = (h[1] == 1);
	return GF2BV_OK;
}

// Does workgroup b of a one-dimensional launch sit on XCD b % 8 on this device?  The gang bulk update keeps a system's multipliers
// in ONE XCD's L2 on that assumption (speed only: results do not depend on it).  Asked once per device with k_probe_xcc -- 64
// workgroups report their XCC_ID: b and b + 8 must agree, 0..7 must differ.  Where it does not hold the gangs take the plain
// (spans, systems) grid of rounds 1-3.  GF2BV_XCD_PIN=1 / 0 skips the question.
int xcd_dispatch_is_round_robin(int device, hipStream_t st, int *ok)
{
	static std::mutex mu;
	static std::map<int, int> known;
	std::lock_guard<std::mutex> lk(mu);
	auto it = known.find(device);
	if (it != known.end()) { *ok = it->second; return GF2BV_OK; }
	int *d = nullptr;
	HIPCHK(pool().alloc((void **)&d, 64 * sizeof(int), device));
	struct Free { int *p; ~Free() { pool().release(p); } } guard{d};
	int h[64];
	int good = 1;
	for (int rep = 0; rep < 2 && good; rep++) {          // (twice: the second launch starts where the first one left the dispatcher)
		k_probe_xcc<<<dim3(64), dim3(64), 0, st>>>(d);
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		for (int b = 0; b < 64; b++) if (h[b] != h[b & 7]) good = 0;
		for (int a = 0; a < 8; a++) for (int b = a + 1; b < 8; b++) if (h[a] == h[b]) good = 0;
	}
	*ok = known[device] = good;
	return GF2BV_OK;
}

// Workgroups of one bulk-update launch (per system).  Each takes one contiguous span of the (tile, row)
// line (see k_update), so the count is free of the tile count: one workgroup per CU, fewer when a span
// would fall under ~4096 rows (a table build costs as much as streaming ~800 rows), shared among the
// systems of a gang.
int pick_update_wgs(i64 est_rows, int ntiles, int nsys, int pinned_wgs = 0)
{
	i64 want = 256;
	want = std::max<i64>(1, want / std::max(1, nsys));
	if (pinned_wgs > 0) want = pinned_wgs;          // XCD-pinned gangs: a system's workgroups fill ITS XCD (32 CUs), one system after the other
	// (a span below ~min_rows rows of a 64-byte column is not worth a workgroup of its own: a table build costs as much
	// as streaming ~1000 of them -- but small passes are latency, not throughput: the chip is mostly idle, so they take
	// as many workgroups as have at least that much to do)
	constexpr i64 min_rows = 2048;
	const i64 cap = std::max<i64>(1, (i64)ntiles * TW / 8 * est_rows / min_rows);
	return (int)std::min(want, cap);
}

// items (tiles, 4-word groups) in [first, end) that this rank owns -- units of 2^ulog items, see owned_item()
i64 owned_count(i64 first, i64 end, int ulog, int world, int wrank)
{
	if (end <= first) return 0;
	if (world <= 1) return end - first;
	i64 n = 0;
	for (i64 u = first >> ulog; u <= (end - 1) >> ulog; u++)
		if (u % world == wrank) n += std::min<i64>(end, (u + 1) << ulog) - std::max<i64>(first, u << ulog);
	return n;
}

// TRSM of block b on the 4-word groups [wlo / 4, end of the row) this rank owns
int launch_trsm(Solver &S, hipStream_t st, int j0, int gb, int wlo, int nw_lo, int nw_hi)
{
	constexpr int WPW = 4;
	const i64 g0 = wlo / WPW, g1 = S.tile_hi * TW / WPW;
	const i64 ng = owned_count(g0, g1, GF2_OWN_LOG - 2, S.world, S.wrank);
	if (ng <= 0) return GF2BV_OK;
	k_block_trsm<TW, WPW><<<dim3((unsigned)ng, S.nsys), dim3(64 * WPW), 0, st>>>(S.M, S.srows, j0, gb, wlo, (int)g0, S.world, S.wrank,
	                                                                           S.panels, S.aux, nw_lo, nw_hi, S.Pc, S.ss());
	HIPCHK(hipGetLastError());
	return GF2BV_OK;
}

// One bulk-update launch of block b on the `ntiles` owned tiles from tile_begin on; `last`: it carries the hand-off
// ("bulk of block b complete") and *handoff receives that event.
int launch_update_timed(Solver &S, hipStream_t st, int b, int j0, int gb, int wlo, u64 *mset, int tile_begin, int ntiles,
                        int nw_lo, bool last, hipEvent_t *handoff)
{
	hipEvent_t ka = nullptr, kb = nullptr;
	if (S.time_kernels) {
		HIPCHK(pool().event(&ka, true)); HIPCHK(pool().event(&kb, true));
		S.kev.push_back(ka); S.kev.push_back(kb);
		if (!S.ext_events) HIPCHK(hipEventRecord(ka, st));
	}
	const i64 est_rows = std::max<i64>(256, S.rows - (i64)j0 * 64);      // alive rows of a dense system (the kernel uses the true bound)
	const bool pin = S.xcd_pin && S.nsys >= 8 && S.nsys % 8 == 0;
	const int wgs = pick_update_wgs(est_rows, ntiles, S.nsys, pin ? S.xcd_wgs : 0);
	hipEvent_t begun = nullptr, done = nullptr;
	if (S.ext_events) { begun = ka; done = S.time_kernels ? kb : (last && !S.flag_sync ? S.evPrio[b] : nullptr); }
	HIPCHK(S.impl->update(dim3((unsigned)wgs, S.nsys), st, S.M, S.rows, S.srows, j0, gb, wlo, S.panels, S.aux, mset,
	                      S.blk_first + b, tile_begin, ntiles, S.world, S.wrank, nw_lo, pin && S.nt_gang,
	                      (const uint4 *)S.Pc, S.ss(), begun, done, pin ? S.nsys : 0));
	if (S.time_kernels && !S.ext_events) HIPCHK(hipEventRecord(kb, st));
	if (!last || S.flag_sync) return GF2BV_OK;
	if (S.ext_events) *handoff = done;
	else { HIPCHK(hipEventRecord(S.evPrio[b], st)); *handoff = S.evPrio[b]; }
	return GF2BV_OK;
}

// Forward elimination: all blocks, no host synchronisation.
//   stream A (panel path): factorise block b on its compact window (G+1 panel steps); then carry
//     block b+1's window forward (k_prio_window: block b's update on those <= 4 words of every row,
//     matrix -> Wb).  Needs the bulk update of block b-1 (stream B) to be complete, nothing newer.
//   stream B (bulk path): block b's TRSM + update on all trailing tiles, as soon as block b is
//     factorised, never writing the next window's words.  So A runs a whole block ahead of B and
//     the two overlap: per-block time is max(panel path, bulk path), and B never idles when it is
//     the longer one.
// ---- one block of the forward elimination, in the three pieces the two streams interleave ----
struct BlockGeom { int j0, gb, wlo, tb, nt_all, gnext; u64 *mset; };
BlockGeom block_geom(const Solver &S, int b)
{
	const int G = S.impl->G;
	BlockGeom g;
	g.j0 = b * G;
	g.gb = std::min(G, S.npanels - g.j0);
	g.wlo = g.j0 + g.gb;
	g.mset = S.mult + (i64)(b % S.nsets) * G * mult_rows(S.rows);
	// trailing tiles; the next block's window [wlo, wlo + gnext) sits in the first one or two of them
	// (two-level: the bulk kernels stop at the outer panel's end, and its last block has no next window to carry forward --
	// the outer pass covers that window like every other tile right of the panel)
	g.tb = g.wlo / TW;
	g.nt_all = (g.wlo < S.wt) ? (int)S.tile_hi - g.tb : 0;
	g.gnext = (b + 1 < S.nblocks && !S.ends_outer_panel(b)) ? std::min(G, S.npanels - g.wlo) : 0;
	return g;
}

// stream A: factorise block b on its compact window -- step s narrows panel s-1 (window half (s-1)&1 -> half s&1)
// while searching panel s; gb+1 launches; evA[b] = "block b factorised"
// fast_only: the block gets the one-launch search and the one-pass narrow step only (see enqueue_forward)
bool fast_block_possible(const Solver &S, const BlockGeom &g)
{
	return S.fast_blocks && g.gb == GF2_GMAX && (i64)(g.j0 + g.gb) * 64 <= S.cols && S.rows >= GF2_FAST_NC;
}

// flag hand-over, panel stream: "block b factorised" (its multipliers are complete) -- and, where the look-ahead of block b
// follows (every block but the last), the wait for the bulk update of block b - 1 that k_prio_window needs
int panel_handover(Solver &S, int b)      // general panel steps: a launch for the announcement (k_narrow_all makes its own)
{
	k_gate<<<dim3(1, S.nsys), dim3(64), 0, S.sA>>>(S.sf, S.st, S.sync_base + b + 1, 0, 0, 0, S.ss());
	HIPCHK(hipGetLastError());
	return GF2BV_OK;
}

int enqueue_block_panel(Solver &S, int b, bool fast_only = false, bool sparse = false)
{
	const BlockGeom g = block_geom(S, b);
	const unsigned row_blocks = (unsigned)((S.rows + 255) / 256);
	u64 *const half[2] = { S.Wb, S.Wb + (i64)GF2_GMAX * S.rows };
	if (fast_only && sparse) {
		// sparse systems: the pool of the search = alive rows with a non-zero window (wmask, left by the look-ahead of the block before)
		if (S.sparse_tier == 0)
			k_block_sparse<256, 4><<<dim3(1, S.nsys), dim3(256), 0, S.sA>>>(S.M, S.rows, S.srows, g.j0, g.gb, 1, b, (const u64 *)half[0], S.st, S.died,
			                                                                 S.panels, S.aux, S.pivcol, S.urow, S.blk_first + b, (const u64 *)S.wmask, S.ss());
		else if (S.sparse_tier == 1)             // (after a give-up, or when the first block counted more rows than the small pool holds: 4096 rows)
			k_block_sparse<512, 8><<<dim3(1, S.nsys), dim3(512), 0, S.sA>>>(S.M, S.rows, S.srows, g.j0, g.gb, 1, b, (const u64 *)half[0], S.st, S.died,
			                                                                S.panels, S.aux, S.pivcol, S.urow, S.blk_first + b, (const u64 *)S.wmask, S.ss());
		else          // (after two: 6144 rows -- 1024 threads at 128 registers, the candidates' words partly in scratch: slow, but still ahead of five general steps)
			k_block_sparse<1024, 6><<<dim3(1, S.nsys), dim3(1024), 0, S.sA>>>(S.M, S.rows, S.srows, g.j0, g.gb, 1, b, (const u64 *)half[0], S.st, S.died,
			                                                                  S.panels, S.aux, S.pivcol, S.urow, S.blk_first + b, (const u64 *)S.wmask, S.ss());
		hipExtLaunchKernelGGL(k_narrow_all, dim3((row_blocks + S.narrow_rpt - 1) / S.narrow_rpt, S.nsys), dim3(256), 0, S.sA, nullptr,
		                      S.ext_events && !S.flag_sync ? S.evA[b] : nullptr, 0, (const u64 *)S.M, S.rows, S.srows, g.j0, b, (const u64 *)half[0],
		                      (const SolveState *)S.st, (const int *)S.died, (const PanelAux *)S.aux, g.mset, S.impl->T, S.narrow_rpt,
		                      S.flag_sync ? DoneSignal{ &S.sf->cnt_narrow, &S.sf->narrow_done, S.sync_base + b + 1 } : DoneSignal{}, S.ss());
		HIPCHK(hipGetLastError());
		if (S.flag_sync) return GF2BV_OK;
		if (!S.ext_events) HIPCHK(hipEventRecord(S.evA[b], S.sA));
		return GF2BV_OK;
	}
	if (fast_only && S.fused_narrow && S.nsys == 1) {
		// search + narrow step in one launch: workgroup 0 searches, the others narrow each panel as soon as it is formed
		// (single systems: a gang is throughput-bound, and its ~130 x nsys narrowing workgroups would hold their LDS for the whole
		// search -- measured equal either way, 243 against 242 systems/s of 32768^2: gangs keep the two launches)
		// at most ~128 narrowing workgroups: they stay resident for the whole search, and a CU that holds TWO of them (2 x 26 KiB
		// of LDS) cannot take a bulk-update workgroup (133 of 160 KiB) until they end -- with 257 workgroups on 256 CUs there is
		// always such a CU, and the pass whose launch lands just behind the search waits ~20 us for its last workgroup
		// (tools/pass_rates.py: every fourth pass of the mid-range of a 65536^2 solve)
		const int rpt = S.fused_rpt > 0 ? S.fused_rpt : (int)std::min<i64>(8, std::max<i64>(S.narrow_rpt, (row_blocks + 127) / 128));
		hipExtLaunchKernelGGL(k_block_fast_narrow, dim3(1 + (row_blocks + rpt - 1) / rpt, S.nsys), dim3(256), 0, S.sA, nullptr,
		                      S.ext_events && !S.flag_sync ? S.evA[b] : nullptr, 0, S.M, S.rows, S.srows, g.j0, g.gb, b, (const u64 *)half[0],
		                      S.st, S.died, S.panels, S.aux, S.pivcol, S.urow, S.blk_first + b, S.Pfast, g.mset, S.impl->T, rpt,
		                      S.flag_sync ? DoneSignal{ &S.sf->cnt_narrow, &S.sf->narrow_done, S.sync_base + b + 1 } : DoneSignal{}, S.ss());
		HIPCHK(hipGetLastError());
		if (S.flag_sync) return GF2BV_OK;       // (the launch announces narrow_done itself)
		if (!S.ext_events) HIPCHK(hipEventRecord(S.evA[b], S.sA));
		return GF2BV_OK;
	}
	if (fast_only) {
		k_block_fast<<<dim3(1, S.nsys), dim3(256), 0, S.sA>>>(S.M, S.rows, S.srows, g.j0, g.gb, 1, b, (const u64 *)half[0], S.st, S.died,
		                                                      S.panels, S.aux, S.pivcol, S.urow, S.blk_first + b, S.Pfast, S.ss());
		hipExtLaunchKernelGGL(k_narrow_all, dim3((row_blocks + S.narrow_rpt - 1) / S.narrow_rpt, S.nsys), dim3(256), 0, S.sA, nullptr,
		                      S.ext_events && !S.flag_sync ? S.evA[b] : nullptr, 0, (const u64 *)S.M, S.rows, S.srows, g.j0, b, (const u64 *)half[0],
		                      (const SolveState *)S.st, (const int *)S.died, (const PanelAux *)S.aux, g.mset, S.impl->T, S.narrow_rpt,
		                      S.flag_sync ? DoneSignal{ &S.sf->cnt_narrow, &S.sf->narrow_done, S.sync_base + b + 1 } : DoneSignal{}, S.ss());
		HIPCHK(hipGetLastError());
		if (S.flag_sync) return GF2BV_OK;       // (the narrow launch announces narrow_done itself)
		if (!S.ext_events) HIPCHK(hipEventRecord(S.evA[b], S.sA));
		return GF2BV_OK;
	}
	// dense blocks: all G panels from a few hundred candidate rows in one launch; the general steps behind it find the
	// block done (or, when it gave up, untouched)
	if (fast_block_possible(S, g))
		k_block_fast<<<dim3(1, S.nsys), dim3(256), 0, S.sA>>>(S.M, S.rows, S.srows, g.j0, g.gb, 0, b, (const u64 *)half[0], S.st, S.died,
		                                                      S.panels, S.aux, S.pivcol, S.urow, S.blk_first + b, S.Pfast, S.ss());
	for (int s = 0; s <= g.gb; s++) {
		const int gp = s - 1, gf = (s < g.gb) ? s : -1;
		const i64 c0 = (i64)(g.j0 + std::max(gf, 0)) * 64;
		const u64 colmask = (S.cols - c0 >= 64) ? ~0ull : ((1ull << (S.cols - c0)) - 1);
		const int find_wgs = gf >= 0 ? (S.units + 3) / 4 : 0;
		const unsigned wgs = (unsigned)find_wgs + (gp >= 0 ? (row_blocks + S.narrow_rpt - 1) / S.narrow_rpt : 0u);
		// the block's last step carries the hand-off event as its own completion signal (no marker packet)
		const bool ext = S.ext_events && !S.flag_sync && s == g.gb && b != S.nblocks - 1;
		hipExtLaunchKernelGGL(k_panel_step, dim3(wgs, S.nsys), dim3(256), 0, S.sA, nullptr, ext ? S.evA[b] : nullptr, 0,
		                      S.M, S.rows, S.srows, g.j0, gp, gf, g.gb, colmask,
		                      (const u64 *)half[s ? (s - 1) & 1 : 0], half[s & 1], S.st, S.died, S.fu, S.units, find_wgs,
		                      S.panels, S.aux, S.pivcol, S.urow, g.mset,
		                      gf == g.gb - 1 ? S.blk_first + b : (int *)nullptr, S.impl->T, S.sparse_mode, S.self_wait, S.narrow_rpt, b, S.ss());
	}
	if (b == S.nblocks - 1)
		k_win_scatter<<<dim3((unsigned)((S.rows * g.gb + 255) / 256), S.nsys), dim3(256), 0, S.sA>>>(S.M, S.rows, S.srows, g.j0, g.gb, half[g.gb & 1], S.died, S.st, S.ss());
	HIPCHK(hipGetLastError());
	if (S.flag_sync) { int rc = panel_handover(S, b); if (rc) return rc; }
	else if (!(S.ext_events && b != S.nblocks - 1)) HIPCHK(hipEventRecord(S.evA[b], S.sA));
	if (S.dbg_sync & 1) HIPCHK(hipDeviceSynchronize());
	return GF2BV_OK;
}

// stream B: TRSM + bulk update of block b on every trailing tile this rank owns (never WRITING the next window);
// waitPrio[b] = "bulk of block b complete"
int enqueue_block_bulk(Solver &S, int b)
{
	const BlockGeom g = block_geom(S, b);
	if (S.bulk_waits_outer) { HIPCHK(hipStreamWaitEvent(S.sB, S.evOuter, 0)); S.bulk_waits_outer = false; }
	if (S.flag_sync) {       // "bulk updates of the blocks before b complete"; wait for block b's multipliers
		// (submitted AFTER the launch that announces narrow_done and BEFORE the panel stream's gate that waits for bulk_done:
		// every wait targets earlier-submitted work.  A gate, not hipStreamWaitValue32: that is a polling kernel as well on this
		// runtime -- __amd_rocclr_streamOpsWait -- but one without a time-out)
		k_gate<<<dim3(1, S.nsys), dim3(64), 0, S.sB>>>(S.sf, S.st, 0, S.sync_base + b, S.sync_base + b + 1, 0, S.ss());
		HIPCHK(hipGetLastError());
	} else HIPCHK(hipStreamWaitEvent(S.sB, S.evA[b], 0));
	bool launched = false;
	if (g.nt_all > 0) {
		int rc = launch_trsm(S, S.sB, g.j0, g.gb, g.wlo, g.wlo, g.wlo + g.gnext);
		if (rc) return rc;
		// 16-byte tiles: the next block's window is whole tiles that are simply left out; when it ends in the middle of a
		// tile (an odd number of window words: the block before a short last one) that tile takes the HALF instance first
		// (the LAST block has no next window: its trailing words start at wlo, possibly in the middle of a tile whose first
		// word is the block's own -- the table entries are zero there, see `keep` in k_update16 -- and nobody else writes it)
		const int wend = g.wlo + g.gnext;
		const int tfull = g.gnext > 0 ? (wend + 1) / 2 : g.wlo / 2;      // first tile with no window word
		const i64 nfull = owned_count(tfull, S.tile_hi, GF2_OWN_LOG - 1, S.world, S.wrank);
		if ((wend & 1) && g.gnext > 0 && owned_count(wend / 2, wend / 2 + 1, GF2_OWN_LOG - 1, S.world, S.wrank) > 0) {
			rc = launch_update_timed(S, S.sB, b, g.j0, g.gb, g.wlo, g.mset, wend / 2, 1, -1, nfull <= 0, &S.waitPrio[b]);
			if (rc) return rc;
			launched = nfull <= 0;
		}
		if (nfull > 0) {
			rc = launch_update_timed(S, S.sB, b, g.j0, g.gb, g.wlo, g.mset, tfull, (int)nfull, 0, true, &S.waitPrio[b]);
			if (rc) return rc;
			launched = true;
		}
	}
	if (!launched && !S.flag_sync) {
		HIPCHK(hipEventRecord(S.evPrio[b], S.sB));      // "bulk of block b complete" (no update launch of this rank carries it)
		S.waitPrio[b] = S.evPrio[b];
	}
	if (S.dbg_sync & 2) HIPCHK(hipDeviceSynchronize());
	return GF2BV_OK;
}

// stream A: the next block's window (needs the bulk update of block b-1, nothing newer)
int enqueue_block_prio(Solver &S, int b)
{
	if (b + 1 >= S.nblocks || S.ends_outer_panel(b)) return GF2BV_OK;
	const BlockGeom g = block_geom(S, b);
	const unsigned row_blocks = (unsigned)((S.rows + 255) / 256);
	if (b > 0 && !S.flag_sync) HIPCHK(hipStreamWaitEvent(S.sA, S.waitPrio[b - 1], 0));
	// the bulk update of block b - 1: announced by the bulk stream's gate of block b (submitted before this one).  A k_gate
	// launch -- ONE spinning wavefront -- and not a wait inside k_prio_window (built in round 3: 32768^2 8.85 -> 8.7 ms): up
	// to thousands of workgroups spinning with 17 KiB of LDS each can sit on every CU before the bulk update they wait for
	// has been placed, which needs 133 KiB of a CU -- a deadlock until the time-out; seen at 327680^2 and beyond.
	if (b > 0 && S.flag_sync) {
		k_gate<<<dim3(1, S.nsys), dim3(64), 0, S.sA>>>(S.sf, S.st, 0, 0, 0, S.sync_base + b, S.ss());
		HIPCHK(hipGetLastError());
	}
	k_prio_window<<<dim3(row_blocks, S.nsys), dim3(256), 0, S.sA>>>(S.M, S.rows, S.srows, g.j0, g.gb, g.wlo, std::max(g.gnext, 1),
	                                                             S.panels, S.aux, g.mset, S.blk_first + b, S.Wb, S.Uwin,
	                                                             S.impl->T, S.st, S.ss(), (const int *)S.died, S.sparse_on ? S.wmask : (u64 *)nullptr);
	HIPCHK(hipGetLastError());
	return GF2BV_OK;
}

int enqueue_forward_begin(Solver &S, bool gather_first_window)
{
	HIPCHK(hipEventRecord(S.ev0, S.sA));
	HIPCHK(hipStreamWaitEvent(S.sB, S.ev0, 0));     // sB starts after the setup memsets on sA
	if (S.npanels > 0 && gather_first_window) {
		const int g0 = std::min(S.impl->G, S.npanels);
		k_win_gather<<<dim3((unsigned)((S.rows * g0 + 255) / 256), S.nsys), dim3(256), 0, S.sA>>>(S.M, S.rows, S.srows, 0, g0, S.Wb, S.st, S.ss());
	}
	return GF2BV_OK;
}

// join: the panel stream waits for the last bulk update; pivot rows get their parked window words
int enqueue_forward_join(Solver &S)
{
	HIPCHK(hipEventRecord(S.ev3, S.sB));
	HIPCHK(hipStreamWaitEvent(S.sA, S.ev3, 0));
	if (S.nblocks > 1 && S.maxr > 0)
		k_unwind<<<dim3((unsigned)((S.maxr * GF2_GMAX + 255) / 256), S.nsys), dim3(256), 0, S.sA>>>(
			S.M, S.srows, S.impl->G, S.npanels, S.nblocks, S.st, S.pivcol, S.urow, S.Uwin, S.world, S.wrank, S.tl_K, S.tl_bend, S.ss());
	HIPCHK(hipGetLastError());
	return GF2BV_OK;
}

int enqueue_check_rhs(Solver &S)
{
	int g = (int)std::min<i64>(1024, (S.rows + 255) / 256);
	k_check_rhs<<<dim3(g, S.nsys), dim3(256), 0, S.sA>>>(S.M, S.rows, S.srows, S.cols, S.died, S.st, S.ss());
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(S.ev1, S.sA));
	return GF2BV_OK;
}

// Two-level elimination, the outer step of the panel of blocks [b0, b1) on the column tiles [t_begin, t_end): pivot rows
// brought up to date there (k_outer_trsm), then all the panel's blocks applied in one pass (k_update16k).  One workgroup per
// ITEM (tile, chunk of rows) -- not persistent ones: CUs come free all the time, and the next panel's elimination, whose
// streams rank above this one, slips its kernels in between (a persistent launch filled every CU until its very end: the
// look-ahead had nowhere to run; reserving CUs for it cost the pass more than the overlap gave back).
// What the outer step of the panel of blocks [b0, b1) needs from its records alone: the row lists and T (P = T x S: the chain
// of panel steps run ONCE, on the identity, one workgroup per source block).  Enqueued on the PANEL stream right behind the
// panel's last block: it runs while the outer stream is still busy with the previous panel's pass (0.26 ms off its path).
int enqueue_outer_prepare(Solver &S, hipStream_t st, int b0, int b1)
{
	const int G = S.impl->G;
	const i64 set_words = (i64)G * mult_rows(S.rows);
	const int npan = (b1 - b0) * G;
	// (lists: two buffers by panel parity, T: two as well -- the previous panel's outer pass may still be reading its own)
	int *gprow = S.oprow + (size_t)((b0 / S.tl_K) % S.nlist) * GF2_OUTER_LISTS;
	u64 *Tm = S.Tm + (size_t)((b0 / S.tl_K) % S.nlist) * ((size_t)GF2_KMAX * GF2_KMAX * GF2_GMAX * 64 * GF2_GMAX);
	k_outer_prow<<<dim3(1, S.nsys), dim3(256), 0, st>>>(b0 * G, b1 - b0, S.panels, S.aux, gprow, S.ss());
	if (!S.outer_chain)
		k_outer_trsm<4, true><<<dim3((unsigned)(npan / 4), S.nsys), dim3(256), 0, st>>>(S.M, S.rows, S.srows, b0 * G, npan, 0, S.panels, S.aux, S.mult,
		                                                                               set_words, b0 % S.nsets, S.nsets, S.impl->T, Tm, S.ss());
	HIPCHK(hipGetLastError());
	return GF2BV_OK;
}

// The outer step of the panel of blocks [b0, b1) on the tiles [t_begin, t_end): the panel's pivot rows P = T x S there (k_outer_apply;
// GF2BV_OUTER_CHAIN=1: the chain itself on every word group), then the pass (k_update16k_wide, one workgroup per item, chunk-major).
int enqueue_outer_apply(Solver &S, hipStream_t st, int b0, int b1, i64 t_begin, i64 t_end)
{
	const int G = S.impl->G;
	const i64 nt = t_end - t_begin;
	if (nt <= 0) return GF2BV_OK;
	const i64 set_words = (i64)G * mult_rows(S.rows);
	int *gprow = S.oprow + (size_t)((b0 / S.tl_K) % S.nlist) * GF2_OUTER_LISTS;
	const int npan = (b1 - b0) * G;
	if (S.outer_chain) {
		const i64 g_begin = t_begin * TW / 4, ng = t_end * TW / 4 - g_begin;
		k_outer_trsm<4, false><<<dim3((unsigned)ng, S.nsys), dim3(256), 0, st>>>(S.M, S.rows, S.srows, b0 * G, npan, (int)g_begin, S.panels, S.aux,
		                                                                          S.mult, set_words, b0 % S.nsets, S.nsets, S.impl->T, (u64 *)nullptr, S.ss());
	} else
		k_outer_apply<<<dim3((unsigned)nt, S.nsys), dim3(512), 0, st>>>(S.M, S.rows, S.srows, b1 - b0, (const int *)gprow,
		                                                                (const u64 *)(S.Tm + (size_t)((b0 / S.tl_K) % S.nlist) * ((size_t)GF2_KMAX * GF2_KMAX * GF2_GMAX * 64 * GF2_GMAX)), (int)t_begin, S.ss());
	HIPCHK(hipGetLastError());
	hipEvent_t ka = nullptr, kb = nullptr;
	if (S.time_kernels) {
		HIPCHK(pool().event(&ka, true)); HIPCHK(pool().event(&kb, true));
		S.kev.push_back(ka); S.kev.push_back(kb);
		if (!S.ext_events) HIPCHK(hipEventRecord(ka, st));
	}
	// items of a dense system: the kernel derives the true count from the alive bound and loops if there are more
	const i64 est_lo = std::min<i64>(S.rows, (i64)b0 * 64 * G) & ~(i64)63, R64 = round_up(S.rows, 64);
	constexpr i64 item_rows = (i64)GF2_WSEG * 1024;
	static_assert((size_t)item_rows * 32 <= kOuterSlackBytes, "the slack behind the matrix covers an item");
	const i64 nch = std::max<i64>(1, (R64 - est_lo + item_rows - 1) / item_rows);
	const i64 wgs = std::min<i64>(nch * nt, (i64)1 << 30);
	hipExtLaunchKernelGGL(k_update16k_wide, dim3((unsigned)wgs, S.nsys), dim3(1024), 0, st, S.ext_events ? ka : nullptr, S.ext_events ? kb : nullptr, 0,
	                      S.M, S.rows, S.srows, b1 - b0, (const int *)gprow, (const u64 *)S.mult,
	                      set_words, b0 % S.nsets, S.nsets, (const int *)(S.blk_first + b0), (const int *)S.died, b1 * G, (int)t_begin, (int)nt, S.ss(), 1);
	HIPCHK(hipGetLastError());
	if (S.time_kernels && !S.ext_events) HIPCHK(hipEventRecord(kb, st));
	if (S.dbg_sync & 2) HIPCHK(hipDeviceSynchronize());
	return GF2BV_OK;
}

int enqueue_window_gather(Solver &S, int b)
{
	const BlockGeom g = block_geom(S, b);
	k_win_gather<<<dim3((unsigned)((S.rows * g.gb + 255) / 256), S.nsys), dim3(256), 0, S.sA>>>(S.M, S.rows, S.srows, g.j0, g.gb, S.Wb, S.st, S.ss());
	HIPCHK(hipGetLastError());
	return GF2BV_OK;
}

int enqueue_forward(Solver &S)
{
	Trace tr;
	int rc = enqueue_forward_begin(S, true);
	if (rc) return rc;
	// Optimistic enqueue for dense systems.  k_block_fast decides ON THE DEVICE whether a block goes the fast way, so the
	// general panel steps have to be enqueued behind it all the same, and on a fast block they are G + 1 empty launches
	// of ~4.5 us each on the critical path (and, beside an outer pass, ~500 workgroups each that take CUs from it).  One look
	// at the device settles it for the usual case: block 0 is enqueued both ways and the host waits for its panel path (a
	// ~50 us stall, once per solve); if the fast search took it, the blocks that can take it at all (full blocks with enough
	// rows left) get k_block_fast + k_narrow_all only.  Should the search give up on one of them after all, it poisons the
	// panel path from there on (SolveState::poison): every later panel-path kernel returns at once, the bulk and outer kernels
	// find no pivots for the unpublished blocks and do nothing, and the host resumes from that block with both paths and the
	// one-level schedule -- the matrix and the window buffer are exactly as that block needs.
	bool optimistic = false;
	SolveState hst{};
	const bool may_probe = S.fast_blocks && S.optimistic && S.nsys == 1 && S.world == 1 && S.nblocks >= 8 && fast_block_possible(S, block_geom(S, 0));
	auto fast_only_ok = [&](int blk) {
		// rows left when the block starts: at most 64 leftover candidates sit below the bound besides the pivots found
		return fast_block_possible(S, block_geom(S, blk)) && S.rows - (i64)blk * 64 * S.impl->G >= GF2_FAST_NC + 128;
	};
	// Sparse systems (round 5): when block 0 did NOT take the dense search, the following full blocks get k_block_sparse + k_narrow_all
	// instead of the G + 1 general panel steps (one-level schedules only; the look-ahead of every block leaves the masks the next
	// block's pool is taken from).  A block it cannot take poisons the panel path like a failed optimistic block: the host resumes
	// there with the general steps for THAT block and goes on sparse behind it (three times at most, then general to the end).
	const bool may_sparse = may_probe && S.sparse_fast && S.tl_K == 0;
	int general_only = -1;                      // (resume) this one block takes the general steps whatever the mode
	auto sparse_ok = [&](int blk) { return S.sparse_on && blk != general_only && fast_block_possible(S, block_geom(S, blk)); };
	// after a poisoned block pb: everything in flight drained, the plan cut back to what has run, the counters rebased
	auto recover = [&](int pb) -> int {
		if (S.sC) HIPCHK(hipStreamSynchronize(S.sC));
		HIPCHK(hipStreamSynchronize(S.sB));
		// (two-level: the outer panels before the poisoned one are complete; the published blocks of the poisoned panel have been
		// applied to its own tiles by the bulk kernels and to everything right of it by its outer pass -- every tile has seen
		// exactly the blocks before pb -- so the rest runs as a one-level schedule)
		if (S.tl_K && pb < S.tl_bend) S.tl_bend = pb / S.tl_K * S.tl_K;
		S.bulk_waits_outer = false;
		HIPCHK(hipMemsetAsync(&S.st->poison, 0, sizeof(int), S.sA));
		S.sync_base += S.nblocks + 1;
		return GF2BV_OK;
	};
	// the first block of each pool size is looked at before anything is enqueued behind it: a pool that cannot serve the system (too
	// many rows carry a bit in a window) costs one block, not a poisoned pipeline of all the blocks behind it
	int sparse_probes = 1;
	auto one_block = [&](int blk) -> int {
		int r;
		bool sp = sparse_ok(blk);
		if ((r = enqueue_block_panel(S, blk, sp || (optimistic && fast_only_ok(blk)), sp))) return r;
		if (sp && sparse_probes > 0) {
			sparse_probes--;
			HIPCHK(hipMemcpyAsync(&hst, S.st, sizeof hst, hipMemcpyDeviceToHost, S.sA));
			HIPCHK(hipStreamSynchronize(S.sA));
			// the pool for the blocks behind this one: a search that counted far more rows with a bit in its window than the middle pool
			// holds goes straight to the largest (MT19937, one bit per output: 4500-5000).  Otherwise the SMALL pool stays while it works,
			// truncated or not -- it is the one that fits beside the bulk update, and the first 1024 candidate rows usually hold a panel's
			// pivots (measured: 17 / 1337 / 137 bits per output count 1100-1560 rows and run 7.4 / 6.3 / 6.7 ms on the small pool, 7.9 /
			// 7.1 / 7.4 on the middle one); a pool that does not gives up and the next size takes over
			const int want = hst.sp_nz > 3400 ? 2 : 0;
			if (hst.poison) {
				if ((r = recover(blk))) return r;
				if (++S.sparse_giveups >= 3) S.sparse_on = false; else sparse_probes = 1;      // (the larger pools get one look each, too)
				S.sparse_tier = std::min(2, std::max(S.sparse_tier + 1, want));
				general_only = blk;
				if ((r = enqueue_block_panel(S, blk, false, false))) return r;
			} else S.sparse_tier = std::max(S.sparse_tier, want);
		}
		if ((r = enqueue_block_bulk(S, blk))) return r;
		if ((r = enqueue_block_prio(S, blk))) return r;
		if (blk == 0 && may_probe) {
			HIPCHK(hipMemcpyAsync(&hst, S.st, sizeof hst, hipMemcpyDeviceToHost, S.sA));
			HIPCHK(hipStreamSynchronize(S.sA));
			optimistic = hst.fast_done == 1;
			if (!optimistic && may_sparse && S.nblocks > 1) {
				// block 1's window is in place (the look-ahead above ran without masks): its masks by a launch of their own
				S.sparse_on = true;
				const BlockGeom g1 = block_geom(S, 1);
				k_window_masks<<<dim3((unsigned)((S.rows + 255) / 256), S.nsys), dim3(256), 0, S.sA>>>((const u64 *)S.Wb, S.rows, g1.gb, (const int *)S.died, S.wmask,
				                                                                                      (const SolveState *)S.st, S.ss());
				HIPCHK(hipGetLastError());
			}
		}
		return GF2BV_OK;
	};
	int b = 0;
	// Two-level part (large single systems, plan_two_level): outer panels of tl_K blocks.  Inside a panel the blocks run
	// exactly as in the one-level loop below -- panel and bulk stream, look-ahead from block to block -- with the bulk kernels
	// confined to the panel's own tiles (tile_hi).  The outer step of panel p goes to a THIRD stream in two parts: first the
	// tiles of panel p + 1 (a small launch), behind which that panel's first window is gathered and its elimination starts,
	// then everything right of them, one workgroup per item so that the next panel's kernels find CUs while it runs.  What
	// the two meet in: the multiplier sets (2 K of them: panel parity), the pivot marks (k_update16k takes "alive behind
	// panel p", not "alive now") and disjoint tiles.
	if (S.tl_K) {
		const int G = S.impl->G;
		hipStream_t so = S.sC ? S.sC : S.sB;           // (GF2BV_SERIAL: one stream, everything in order)
		bool big_recorded = false;                     // evBig holds the end of the previous panel's pass (side launches)
		for (int p0 = 0; p0 < S.tl_bend; p0 += S.tl_K) {
			const int p1 = p0 + S.tl_K;
			if (p0 > 0) {
				HIPCHK(hipStreamWaitEvent(S.sA, S.evPri, 0));
				if ((rc = enqueue_window_gather(S, p0))) return rc;
			}
			S.tile_hi = (i64)p1 * G / TW;
			for (b = p0; b < p1; b++)
				if ((rc = one_block(b))) return rc;
			S.tile_hi = S.ntiles;
			// the outer step needs the panel's records and multipliers (panel stream), not its bulk updates (other tiles)
			if ((rc = enqueue_outer_prepare(S, S.sA, p0, p1))) return rc;
			HIPCHK(hipEventRecord(S.evPanelDone, S.sA));
			HIPCHK(hipStreamWaitEvent(so, S.evPanelDone, 0));
			const i64 t_out = S.ntiles;
			const i64 t0 = (i64)p1 * G / TW, t1 = std::min<i64>(t_out, t0 + (i64)S.tl_K * G / TW);
			// (late round 5) the outer step on the NEXT panel's tiles -- a short launch on an underused chip, ~0.3 ms -- goes to the
			// inner elimination's bulk stream, idle at this point, and runs BESIDE the start of the pass proper instead of before it:
			// it needs the previous pass complete (evBig; it used to follow it in stream order) and this panel's T; the pass proper
			// needs T alone (disjoint tiles).  No new stream.
			const bool side = S.outer_side && S.sB != so && S.sB != S.sA;
			if (side) {
				HIPCHK(hipStreamWaitEvent(S.sB, S.evPanelDone, 0));
				if (big_recorded) HIPCHK(hipStreamWaitEvent(S.sB, S.evBig, 0));
				if ((rc = enqueue_outer_apply(S, S.sB, p0, p1, t0, t1))) return rc;
				HIPCHK(hipEventRecord(S.evPri, S.sB));
				if ((rc = enqueue_outer_apply(S, so, p0, p1, t1, t_out))) return rc;
			} else {
				if ((rc = enqueue_outer_apply(S, so, p0, p1, t0, t1))) return rc;
				HIPCHK(hipEventRecord(S.evPri, so));
				if ((rc = enqueue_outer_apply(S, so, p0, p1, t1, t_out))) return rc;
			}
			// (round 6, VERDICT item 2a: P = T x S of the NEXT panel started behind the first chunk of this pass, beside its rest -- built,
			// bit-exact, no gain: the launch is LDS table work like the pass itself, not idle chip time.  profiles/r06_outer_early.txt)
			HIPCHK(hipEventRecord(S.evBig, so));
			big_recorded = true;
		}
		HIPCHK(hipEventRecord(S.evOuter, so));
		b = S.tl_bend;
		HIPCHK(hipStreamWaitEvent(S.sA, S.evPri, 0));
		if (b < S.nblocks && (rc = enqueue_window_gather(S, b))) return rc;
		S.bulk_waits_outer = S.sC != nullptr;          // the one-level bulk updates behind it touch every trailing tile
	}
	for (; b < S.nblocks; b++)
		if ((rc = one_block(b))) return rc;
	if ((rc = enqueue_forward_join(S))) return rc;
	tr.mark("forward: all blocks submitted");          // (host side only: the device is still working; what follows waits for it)
	while (optimistic || S.sparse_on) {
		HIPCHK(hipMemcpyAsync(&hst, S.st, sizeof hst, hipMemcpyDeviceToHost, S.sA));
		HIPCHK(hipStreamSynchronize(S.sA));
		if (!hst.poison) break;
		{                                               // a block the fast search could not take: resume there, both paths
			const int pb = hst.poison - 1;
			if ((rc = recover(pb))) return rc;
			optimistic = false;
			if (S.sparse_on && ++S.sparse_giveups > 3) S.sparse_on = false;
			S.sparse_tier = std::min(2, S.sparse_tier + 1);
			general_only = pb;
			// (the poisoned block's window is in place; the blocks behind it take their masks from the look-aheads again)
			for (b = pb; b < S.nblocks; b++)
				if ((rc = one_block(b))) return rc;
			if ((rc = enqueue_forward_join(S))) return rc;
		}
	}
	return enqueue_check_rhs(S);
}

// back-substitution on Y = selected columns of U (RHS [+ free columns]); then scatter.
int enqueue_backward(Solver &S, const std::vector<int> &ycols_host)
{
	S.ny = (int)ycols_host.size();
	const i64 nyw = (S.ny + 63) / 64;
	S.ys = round_up(nyw, YTW);
	HIPCHK(pool().alloc((void **)&S.ycols, sizeof(int) * S.ny, S.device));
	HIPCHK(hipMemcpyAsync(S.ycols, ycols_host.data(), sizeof(int) * S.ny, hipMemcpyHostToDevice, S.sA));
	HIPCHK(pool().alloc((void **)&S.Y, sizeof(u64) * std::max<i64>(1, S.maxr) * S.ys, S.device));
	HIPCHK(hipMemsetAsync(S.Y, 0, sizeof(u64) * std::max<i64>(1, S.maxr) * S.ys, S.sA));
	HIPCHK(pool().alloc((void **)&S.out, sizeof(u64) * S.ny * std::max<i64>(1, S.cw), S.device));
	HIPCHK(hipMemsetAsync(S.out, 0, sizeof(u64) * S.ny * std::max<i64>(1, S.cw), S.sA));
	u64 *bmult = S.mult;              // forward multipliers are dead by now
	const int rpb = 2048;
	if (S.maxr > 0) {
		// (a launch dimension holds fewer than 2^32 work-items: pivots go in chunks of 2^24 wavefronts)
		const i64 kchunk = std::max<i64>(1, (1ll << 24) / nyw);
		for (i64 k0 = 0; k0 < S.maxr; k0 += kchunk) {
			const i64 waves = std::min(kchunk, S.maxr - k0) * nyw;
			k_extract_y<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, S.sA>>>(S.M, S.srows, S.st, S.urow, S.pivcol,
			                                                                      S.ycols, S.ny, S.Y, S.ys, k0);
		}
		const int ytiles = (int)(S.ys / YTW);
		for (int q = S.npanels - 1; q >= 1; q--) {
			const i64 bound = std::min<i64>((i64)64 * q, S.maxr);      // pivots before panel q
			int g = (int)std::min<i64>(1024, (bound + 255) / 256);
			k_gather_mult_u<<<dim3(g), dim3(256), 0, S.sA>>>(S.M, S.srows, q, S.panels + q, S.urow, bmult);
			const i64 nrb = (bound + rpb - 1) / rpb;
			HIPCHK(launch_ysweep(dim3((unsigned)(ytiles * nrb)), S.sA, S.Y, S.ys, S.maxr, S.panels + q, bmult, ytiles, rpb));
		}
		for (i64 k0 = 0; k0 < S.maxr; k0 += kchunk << 6) {
			const i64 thr = std::min(kchunk << 6, S.maxr - k0) * nyw;
			k_scatter_solution<<<dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, S.sA>>>(S.Y, S.ys, S.st, S.pivcol, S.ny,
			                                                                              S.out, std::max<i64>(1, S.cw), k0);
		}
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(S.ev2, S.sA));
	return GF2BV_OK;
}

// Blocked parity back-substitution straight into the solution words (no Y matrix), for up to GF2_BSV
// right-hand sides at once: solve_one (the RHS column) and small kernel bases (the free columns + RHS).
// U is streamed once for all of them.
int enqueue_backward_parity(Solver &S, const std::vector<int> &ycols_host)
{
	S.ny = (int)ycols_host.size();
	const i64 cw = std::max<i64>(1, S.cw);
	HIPCHK(pool().alloc((void **)&S.ycols, sizeof(int) * S.ny, S.device));
	HIPCHK(hipMemcpyAsync(S.ycols, ycols_host.data(), sizeof(int) * S.ny, hipMemcpyHostToDevice, S.sA));
	HIPCHK(pool().alloc((void **)&S.out, sizeof(u64) * S.ny * cw, S.device));
	HIPCHK(hipMemsetAsync(S.out, 0, sizeof(u64) * S.ny * cw, S.sA));
	unsigned char *accv = reinterpret_cast<unsigned char *>(S.mult);     // forward multipliers are dead by now (2*G*8 bytes per row)
	const i64 nacc = std::max<i64>(1, S.maxr);
	// the diagonal blocks of U, gathered by the whole chip into compact form for the serial walk of k_bs_near
	HIPCHK(pool().alloc((void **)&S.Y, sizeof(u64) * 64 * 16 * std::max(1, S.npanels), S.device));
	if (S.npanels > 0)
		k_bs_diag<<<dim3(S.npanels), dim3(256), 0, S.sA>>>(S.M, S.srows, S.npanels, S.panels, S.urow, S.Y, SysStride{0, 0}, (i64)0);
	// Round 4: from sixteen groups (16384 columns) up the diagonal blocks are inverted up front (k_bs_inv, all groups in one launch, ~90 us) and the
	// serial walk of a link (k_bs_near, 12 us) becomes a matrix-vector product (k_bs_near2, ~4 us).  GF2BV_BS_INV=0 / 1: never / always.
	const int ngroups = (S.npanels + GF2_BSG - 1) / GF2_BSG;
	bool inv = ngroups >= 16;
	if (const char *e = getenv("GF2BV_BS_INV")) inv = atoi(e) != 0 && ngroups >= 1;
	if (inv) {
		HIPCHK(pool().alloc((void **)&S.Minv, sizeof(u64) * (size_t)ngroups * 16 * 64 * 16, S.device));
		static std::mutex mu;
		static std::map<int, bool> raised;          // device -> the > 64 KiB dynamic-LDS attribute has been raised there
		{
			std::lock_guard<std::mutex> lk(mu);
			if (!raised[S.device]) {
				HIPCHK(hipFuncSetAttribute((const void *)k_bs_inv, hipFuncAttributeMaxDynamicSharedMemorySize, GF2_BSINV_LDS));
				raised[S.device] = true;
			}
		}
		k_bs_inv<<<dim3(ngroups), dim3(1024), GF2_BSINV_LDS, S.sA>>>(S.Y, S.npanels, S.panels, S.pivcol, S.Minv);
	}
	// right-hand sides in groups of GF2_BSV (U is streamed once per group; the groups are independent)
	for (int v0 = 0; v0 < S.ny; v0 += GF2_BSV) {
		const int nv = std::min(GF2_BSV, S.ny - v0);
		for (int qb = S.npanels; qb > 0; qb -= GF2_BSG) {
			const int qa = std::max(0, qb - GF2_BSG);
			const int waves = (qb - qa) * 64;
			k_bs_far<<<dim3((waves + 3) / 4), dim3(256), 0, S.sA>>>(S.M, S.srows, cw, qa, qb, S.panels, S.urow, S.pivcol, S.ycols + v0, nv,
			                                                        S.out + (i64)v0 * cw, accv, nacc, SysStride{0, 0}, (i64)0);
			if (inv)
				k_bs_near2<<<dim3(1), dim3(1024), 0, S.sA>>>(cw, qa, qb, S.panels, S.pivcol, nv, S.out + (i64)v0 * cw, accv, nacc, S.Minv,
				                                             (S.npanels - qb) / GF2_BSG);
			else
				k_bs_near<<<dim3(1), dim3(256), 0, S.sA>>>(S.Y, cw, qa, qb, S.panels, S.pivcol, nv, S.out + (i64)v0 * cw, accv, nacc,
				                                           SysStride{0, 0}, (i64)0, (i64)0);
		}
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(S.ev2, S.sA));
	return GF2BV_OK;
}

// solve_one of a whole gang: the same three kernels with blockIdx.y = system -- one chain of 2 x ceil(npanels / 16) launches
// for the gang instead of one per system (a 32768^2 system's chain is 64 launches of 6 + 13 us: 1.2 ms during which one
// workgroup walks a diagonal block; 24 of them in a row were 15 % of a gang's wall time, profiles/r04_batch_budget.txt).
// The gang owns ycols / out / Y; system s finds its solution at out + s * cw (the views point there, own_bs = false).
int enqueue_backward_gang(Solver &S)
{
	S.ny = 1;
	const i64 cw = std::max<i64>(1, S.cw);
	S.h_ycol = (int)S.cols;
	HIPCHK(pool().alloc((void **)&S.ycols, sizeof(int), S.device));
	HIPCHK(hipMemcpyAsync(S.ycols, &S.h_ycol, sizeof(int), hipMemcpyHostToDevice, S.sA));
	HIPCHK(pool().alloc((void **)&S.out, sizeof(u64) * cw * S.nsys, S.device));
	HIPCHK(hipMemsetAsync(S.out, 0, sizeof(u64) * cw * S.nsys, S.sA));
	unsigned char *accv = reinterpret_cast<unsigned char *>(S.mult);     // forward multipliers are dead by now
	const i64 nacc = std::max<i64>(1, S.maxr);
	const i64 dgw = (i64)64 * 16 * std::max(1, S.npanels);
	HIPCHK(pool().alloc((void **)&S.Y, sizeof(u64) * dgw * S.nsys, S.device));
	if (S.npanels > 0)
		k_bs_diag<<<dim3(S.npanels, S.nsys), dim3(256), 0, S.sA>>>(S.M, S.srows, S.npanels, S.panels, S.urow, S.Y, S.ss(), dgw);
	for (int qb = S.npanels; qb > 0; qb -= GF2_BSG) {
		const int qa = std::max(0, qb - GF2_BSG);
		const int waves = (qb - qa) * 64;
		k_bs_far<<<dim3((waves + 3) / 4, S.nsys), dim3(256), 0, S.sA>>>(S.M, S.srows, cw, qa, qb, S.panels, S.urow, S.pivcol, S.ycols, 1,
		                                                                S.out, accv, nacc, S.ss(), cw);
		k_bs_near<<<dim3(1, S.nsys), dim3(256), 0, S.sA>>>(S.Y, cw, qa, qb, S.panels, S.pivcol, 1, S.out, accv, nacc, S.ss(), dgw, cw);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(S.ev2, S.sA));
	return GF2BV_OK;
}

int enqueue_backward_single(Solver &S)
{
	return enqueue_backward_parity(S, std::vector<int>(1, (int)S.cols));
}

int solver_enqueue(Solver &S)
{
	Trace tr;
	int rc = solver_alloc(S);
	tr.mark("solver_alloc");
	if (rc) return rc;
	rc = enqueue_forward(S);
	tr.mark("enqueue_forward");
	if (rc) return rc;
	if (S.mode == GF2BV_MODE_SINGLE) {
		if (getenv("GF2BV_YSWEEP")) {                  // the general multi-RHS path, for cross-checking
			std::vector<int> yc(1, (int)S.cols);
			return enqueue_backward(S, yc);
		}
		return enqueue_backward_single(S);
	}
	return GF2BV_OK;
}

// Export, first half: (mode 1: rank and pivot columns to the host, kernel-basis back-substitution,)
// then the asynchronous device-to-host copies of everything the result needs.
int finish_begin(Solver &S)
{
	HIPCHK(pool().event(&S.evx, true));
	if (S.mode == GF2BV_MODE_AFFINE_SPACE) {
		// the kernel basis needs rank and pivot columns on the host (one sync) to lay out
		// the free columns in M4RI's order (_internal.c:348)
		HIPCHK(hipMemcpyAsync(&S.hst, S.st, sizeof S.hst, hipMemcpyDeviceToHost, S.sA));
		HIPCHK(hipStreamSynchronize(S.sA));
		std::vector<int32_t> piv(S.hst.rank);
		if (S.hst.rank)
			HIPCHK(hipMemcpy(piv.data(), S.pivcol, sizeof(int) * S.hst.rank, hipMemcpyDeviceToHost));
		std::vector<int> order(S.cols);
		for (i64 i = 0; i < S.cols; i++) order[i] = (int)i;
		for (int i = 0; i < S.hst.rank; i++) std::swap(order[i], order[piv[i]]);
		S.free_order.assign(order.begin() + S.hst.rank, order.end());
		std::vector<int> yc;
		if (!S.hst.inconsistent) yc = S.free_order;
		yc.push_back((int)S.cols);
		// up to GF2_BS_MAXRHS right-hand sides (kernel dimensions of the reference's use: solve_all caps at 16): the parity
		// path, GF2_BSV of them per pass over U; larger bases: the table sweeps over the bit matrix Y (one pass over Y per
		// panel whatever the dimension)
		int rc = ((int)yc.size() <= GF2_BS_MAXRHS && !getenv("GF2BV_YSWEEP")) ? enqueue_backward_parity(S, yc) : enqueue_backward(S, yc);
		if (rc) return rc;
	}
	HIPCHK(hipMemcpyAsync(&S.hst, S.st, sizeof S.hst, hipMemcpyDeviceToHost, S.sA));
	S.hout.resize((size_t)S.ny * std::max<i64>(1, S.cw));
	HIPCHK(hipMemcpyAsync(S.hout.data(), S.out, sizeof(u64) * S.hout.size(), hipMemcpyDeviceToHost, S.sA));
	S.hp.resize(std::max(1, S.npanels));
	HIPCHK(hipMemcpyAsync(S.hp.data(), S.panels, sizeof(PanelRec) * S.hp.size(), hipMemcpyDeviceToHost, S.sA));
	S.hpiv.resize(std::max<i64>(1, S.maxr));
	HIPCHK(hipMemcpyAsync(S.hpiv.data(), S.pivcol, sizeof(int) * S.hpiv.size(), hipMemcpyDeviceToHost, S.sA));
	HIPCHK(hipEventRecord(S.evx, S.sA));
	return GF2BV_OK;
}

// Export, second half: wait for the copies and build the result object.
int finish_end(Solver &S, gf2bv_result **out)
{
	Trace tr;
	HIPCHK(hipStreamSynchronize(S.sA));
	HIPCHK(hipStreamSynchronize(S.sB));
	tr.mark("finish: sync");
	const SolveState &hst = S.hst;
	if (hst.gate_timeout) {
		if (!S.flag_sync) return fail(GF2BV_ERR_HIP, "a stream hand-over gate timed out on the device");
		forget_concurrency(S.device);
		return GF2BV_RETRY_EVENTS;
	}
	const std::vector<u64> &hout = S.hout;
	const std::vector<PanelRec> &hp = S.hp;
	S.hpiv.resize(hst.rank);
	const std::vector<int32_t> &piv = S.hpiv;
	hipEvent_t evx = S.evx;

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
	st.panels_per_sweep = S.impl->G;
	st.tables_per_sweep = S.impl->G * S.impl->T;
	st.table_bits = (64 + S.impl->T - 1) / S.impl->T;          // (16-byte tiles: 8 byte fields per panel)
	st.tile_words = TW;
	st.gang_systems = S.view ? S.gang_nsys : S.nsys;
	st.search_handovers = hst.self_giveups;
	st.fast_blocks = hst.fast_blocks;
	{
		const int G = S.impl->G;
		for (int b = 0; b < S.nblocks; b++) {
			const int j0 = b * G, gb = std::min(G, S.npanels - j0), wlo = j0 + gb;
			int pend = 0, any = 0;
			for (int g = 0; g < gb; g++) { any |= hp[j0 + g].p; pend = hp[j0 + g].start + hp[j0 + g].p; }
			if (!any || wlo >= S.wt) continue;
			const double rows_swept = (double)(S.rows - pend);
			st.n_sweeps++;
			st.sweep_words += rows_swept * (double)(S.wt - wlo);
			st.row_xors += rows_swept * (double)(S.impl->T * gb);
			// what the launches moved: a block inside an outer panel is applied by k_update16 up to the panel's end only, the
			// rest of the row takes the whole panel in one outer pass (counted at the panel's last block, two launches)
			if (S.tl_K && b < S.tl_bend) {
				const int pend_w = (b / S.tl_K + 1) * S.tl_K * G;
				const i64 out_w = S.wt;
				st.outer_blocks++;
				if (pend_w > wlo) { st.hbm_words += rows_swept * (double)(std::min<i64>(pend_w, S.wt) - wlo); st.bulk_launches++; }
				if ((b + 1) % S.tl_K == 0 && out_w > pend_w) { st.hbm_words += rows_swept * (double)(out_w - pend_w); st.bulk_launches += 2; }      // (the next panel's tiles, then the rest -- in two halves on two streams since late round 5)
			} else { st.hbm_words += rows_swept * (double)(S.wt - wlo); st.bulk_launches++; }
		}
	}
	st.ms_pack = S.ms_pack;
	st.handover_retries = g_attempt;
	(void)hipEventElapsedTime(&st.ms_eliminate, S.ev0, S.ev1);
	(void)hipEventElapsedTime(&st.ms_backsub, S.ev1, S.ev2);
	(void)hipEventElapsedTime(&st.ms_export, S.ev2, evx);
	{
		// ms_sweep = the time during which AT LEAST ONE bulk-update launch was running: launches of one stream add up as before
		// (every one-level plan), launches that run side by side -- the two halves of an outer pass on their two streams, the next
		// panel's inner updates beside them -- count once (summed, the split pass of late round 5 would be counted twice)
		std::vector<std::pair<float, float>> iv;
		for (size_t i = 0; i + 1 < S.kev.size(); i += 2) {
			float ms = 0, at = 0;
			(void)hipEventElapsedTime(&ms, S.kev[i], S.kev[i + 1]);
			if (hipEventElapsedTime(&at, S.ev0, S.kev[i]) != hipSuccess) { (void)hipGetLastError(); at = iv.empty() ? 0.f : iv.back().second; }
			iv.push_back({ at, at + ms });
		}
		std::sort(iv.begin(), iv.end());
		float lo = 0, hi = -1;
		for (const auto &v : iv) {
			if (hi < 0) { lo = v.first; hi = v.second; }
			else if (v.first <= hi) hi = std::max(hi, v.second);
			else { st.ms_sweep += hi - lo; lo = v.first; hi = v.second; }
		}
		if (hi >= 0) st.ms_sweep += hi - lo;
	}
	st.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - S.t_begin).count();
	tr.mark("finish: result");
	*out = R;
	return GF2BV_OK;
}

int solver_finish(Solver &S, gf2bv_result **out)
{
	int rc = finish_begin(S);
	if (rc) return rc;
	return finish_end(S, out);
}

// A non-owning window on system s of a gang: same streams and events, pointers moved to that
// system's matrix and arena.  Back-substitution and export then run per system, unchanged.
int make_view(const Solver &S, int s, Solver &V)
{
	V = S;
	V.view = true;
	V.gang_nsys = S.nsys;
	V.nsys = 1; V.m_stride = 0; V.arena_stride = 0; V.src_sys_words = 0;
	V.own_sA = V.own_sB = false;
	V.src = nullptr; V.tmp_src = nullptr;
	V.kev.clear(); V.evA.clear(); V.evPrio.clear();
	const i64 ao = (i64)S.arena_stride * s;
	auto mv = [ao](auto *&p) { p = reinterpret_cast<decltype(+p)>(reinterpret_cast<char *>(p) + ao); };
	V.M = S.M + S.m_stride * s;
	V.arena = (char *)S.arena + ao;
	mv(V.st); mv(V.panels); mv(V.aux); mv(V.fu); mv(V.died); mv(V.pivcol); mv(V.urow); mv(V.blk_first); mv(V.mult); mv(V.Wb); mv(V.Uwin); mv(V.Pfast); if (V.Pc) mv(V.Pc);
	V.Y = nullptr; V.ycols = nullptr; V.out = nullptr; V.Minv = nullptr;
	V.ev2 = nullptr;
	HIPCHK(pool().event(&V.ev2, true));
	return GF2BV_OK;
}

// One gang: S.nsys same-shape systems, forward elimination in lock-step (one set of launches),
// then back-substitution and export system by system.
int solve_gang(Solver &S, gf2bv_result **out)
{
	int rc = solver_alloc(S);
	if (rc) return rc;
	rc = enqueue_forward(S);
	if (rc) return rc;
	std::vector<Solver> V(S.nsys);
	for (int s = 0; s < S.nsys; s++) {
		V[s].device = S.device;
		rc = make_view(S, s, V[s]);
		if (rc) return rc;
	}
	if (S.mode == GF2BV_MODE_SINGLE) {
		const bool per_system = !S.gang_bs;      // (GF2BV_GANG_BS=0, A/B: one chain per system, as rounds 1-3)
		if (per_system)
			for (int s = 0; s < S.nsys; s++) {
				rc = enqueue_backward_single(V[s]);
				if (rc) return rc;
			}
		else {
			rc = enqueue_backward_gang(S);
			if (rc) return rc;
			for (int s = 0; s < S.nsys; s++) {
				V[s].out = S.out + (i64)s * std::max<i64>(1, S.cw); V[s].ny = 1; V[s].own_bs = false;
				HIPCHK(hipEventRecord(V[s].ev2, S.sA));
			}
		}
	}
	for (int s = 0; s < S.nsys; s++) {
		rc = finish_begin(V[s]);
		if (rc) return rc;
	}
	for (int s = 0; s < S.nsys; s++) {
		rc = finish_end(V[s], &out[s]);
		if (rc) return rc;
	}
	if (S.time_kernels) {        // one set of bulk-update launches served the whole gang: every member reports the gang's time
		float total = 0;
		for (size_t i = 0; i + 1 < S.kev.size(); i += 2) {
			float ms = 0;
			(void)hipEventElapsedTime(&ms, S.kev[i], S.kev[i + 1]);
			total += ms;
		}
		for (int s = 0; s < S.nsys; s++) out[s]->stats.ms_sweep = total;
	}
	return GF2BV_OK;
}

// Gang size for nsys same-shape systems (free_b: free device memory, < 0 = ask the current device).
i64 pick_gang(i64 nsys, i64 rows, i64 cols, i64 free_bytes = -1)
{
	// gang size: ~4.5 GiB of working matrices per gang (32768^2: 32 systems -- round 4: 192 x 32768^2 run 3.22 ms per system in gangs
	// of 32 against 3.22-3.32 in gangs of 24, and 512 systems are 16 equal gangs instead of 21 + a part gang; 4096^2: 64), at
	// least four gangs (round 3, 144 x
	// 32768^2 on one box: gangs of 4 / 8 / 12 / 18 / 24 / 36 -> 231 / 248 / 250 / 239-254 / 255-256 / 255 systems per second)
	// when there are enough systems (measured on MI355X, 48 x 32768^2: gang 4/8/16 -> 7.6/6.3/6.0 ms per
	// system, one system at a time 15-28; 64 x 4096^2: 0.22 ms per system against 1.8)
	const double per_sys = 1.05 * 8.0 * (double)(rows + 64) * (double)((cols + 64) / 64 + TW + 4 * GF2_GMAX);
	i64 gang = std::max<i64>(2, std::min<i64>(64, (i64)(4.5 * 1073741824.0 / per_sys)));
	// (at least TWO gangs -- one per host thread -- where round 3 asked for four: with a system per XCD larger gangs win; a rank's share of
	// the configs[3] job at 8 GPUs, 64 x 32768^2: 2 x 32 run 3.25 ms per system against 3.41 for 4 x 16, profiles/r05_batch_scans.txt)
	gang = std::min(gang, std::max<i64>(1, (nsys + 1) / 2));
	// equal gangs, an even number of them (two host threads take alternate gangs): 64 systems of 32768^2 go as 4 x 16
	// (3.90 ms per system) rather than 4 x 14 + 8 (4.10)
	if (gang < nsys) {
		i64 ngangs = (nsys + gang - 1) / gang;
		if (ngangs > 1 && (ngangs & 1)) ngangs++;
		gang = (nsys + ngangs - 1) / ngangs;
	}
	// whole multiples of 8 systems: the bulk update of such a gang keeps every system on ONE XCD (k_update16: xcd_nsys)
	if (gang >= 8) gang = gang / 8 * 8;
	if (const char *e = getenv("GF2BV_GANG")) { int v = atoi(e); if (v >= 1) gang = v; }
	gang = std::max<i64>(1, std::min<i64>(gang, nsys));
	{
		size_t free_b = 0, total_b = 0;
		if (free_bytes >= 0) gang = std::max<i64>(1, std::min<i64>(gang, (i64)(0.4 * (double)free_bytes / per_sys)));
		else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
			gang = std::max<i64>(1, std::min<i64>(gang, (i64)(0.4 * (double)free_b / per_sys)));
	}
	return gang;
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

// ---- small systems: the whole solve in one launch (k_small_solve) ---------------------------------------------------
// Eligible: the augmented matrix + two sets of nibble tables fit the LDS of one workgroup ((rows + 512) x odd row pitch <= 18432 words = 144 KiB, cols <= 1023,
// rows <= 4096).  Per call: ONE host-to-device copy out of a pinned staging buffer (offsets + digits, or the packed words;
// none when the matrix already lives on the device), one launch, and the result written by the kernel straight into pinned
// host memory -- no device-to-host copy.  GF2BV_SMALL=0 sends these systems down the blocked path instead (tests compare both).
struct SmallStage {            // pinned staging of one call in flight (input copy / zero-copy source, and the result)
	int device = -1;
	char *h_in = nullptr, *h_out = nullptr;
	size_t in_cap = 0, out_cap = 0;
};
// Stages are handed out per call and go back afterwards: as many exist as calls were ever in flight at once on a device (a
// thread_local pair per host thread would pin 320 KiB for every thread a pool ever creates).  Never freed at exit: the HIP runtime
// may be gone by then.
struct StagePool {
	std::mutex mu;
	std::map<int, std::vector<SmallStage *>> idle;
	SmallStage *take(int device)
	{
		std::lock_guard<std::mutex> lk(mu);
		auto &v = idle[device];
		if (!v.empty()) { SmallStage *s = v.back(); v.pop_back(); return s; }
		SmallStage *s = new SmallStage();
		s->device = device;
		return s;
	}
	void give(SmallStage *s) { std::lock_guard<std::mutex> lk(mu); idle[s->device].push_back(s); }
};
StagePool &stage_pool() { static StagePool *p = new StagePool(); return *p; }

bool small_eligible(i64 rows, i64 cols)
{
	if (const char *e = getenv("GF2BV_SMALL")) if (atoi(e) == 0) return false;
	const i64 wt = (cols + 1 + 63) / 64;
	return cols <= GF2_SMALL_MAXCOLS && rows <= GF2_SMALL_MAXROWS && (rows + 512) * small_pitch(wt) <= GF2_SMALL_LDS_WORDS;
}

int small_stage(SmallStage &G, size_t in_bytes, size_t out_bytes)
{
	if (in_bytes > G.in_cap) {
		if (G.h_in) (void)hipHostFree(G.h_in);
		G.h_in = nullptr; G.in_cap = 0;
		const size_t cap = std::max<size_t>(in_bytes * 2, (size_t)256 << 10);
		HIPCHK(hipHostMalloc((void **)&G.h_in, cap, hipHostMallocDefault));
		G.in_cap = cap;
	}
	if (out_bytes > G.out_cap) {
		if (G.h_out) (void)hipHostFree(G.h_out);
		G.h_out = nullptr; G.out_cap = 0;
		const size_t cap = std::max<size_t>(out_bytes * 2, (size_t)64 << 10);
		HIPCHK(hipHostMalloc((void **)&G.h_out, cap, hipHostMallocDefault));
		G.out_cap = cap;
	}
	return GF2BV_OK;
}

// One of: d_words (device, row-major, d_stride) | h_words (host, row-major, h_stride) | h_digits + h_off (host, bpd bits per digit)
struct SmallInput {
	const u64 *d_words = nullptr; i64 d_stride = 0;
	const u64 *h_words = nullptr; i64 h_stride = 0;
	const uint32_t *h_digits = nullptr; const i64 *h_off = nullptr; int bpd = 0;
};

int small_solve(const SmallInput &in, i64 rows, i64 cols, int mode, int device, hipStream_t stream, gf2bv_result **out)
{
	const auto t_begin = std::chrono::steady_clock::now();
	const i64 wt = (cols + 1 + 63) / 64, cw = (cols + 63) / 64;
	const bool want_basis = mode == GF2BV_MODE_AFFINE_SPACE;
	const size_t head = 16 + 4 * (size_t)((cols + 1) & ~(i64)1);
	const size_t res_bytes = head + 8 * (size_t)cw * (1 + (want_basis ? (size_t)cols : 0));
	const size_t out_bytes = res_bytes + 64;                 // (+ 8 phase counters of the probe, GF2BV_SMALL_PROBE=1)
	const bool probe = getenv("GF2BV_SMALL_PROBE") != nullptr;
	size_t in_bytes = 0;
	i64 ndig = 0;
	if (in.h_digits) { ndig = in.h_off[rows]; in_bytes = sizeof(i64) * (size_t)(rows + 1) + sizeof(uint32_t) * (size_t)std::max<i64>(ndig, 1); }
	else if (in.h_words) in_bytes = sizeof(u64) * (size_t)(rows * wt);
	struct Stage { SmallStage *s; ~Stage() { stage_pool().give(s); } } stage{ stage_pool().take(device) };
	SmallStage &G = *stage.s;
	int rc = small_stage(G, in_bytes, out_bytes);
	if (rc) return rc;
	// Stream: a matrix that already lives on the device is whatever `stream` (NULL = the null stream) has produced -- the launch goes
	// onto that very stream, as on the blocked path.  Host inputs depend on nothing: a pool stream of their own (non-blocking).
	hipStream_t st = stream;
	bool own_stream = false;
	if (!st && !in.d_words) { HIPCHK(pool().stream(&st, device, false)); own_stream = true; }
	struct Back { hipStream_t st; int device; bool own; void *d_in; ~Back() { if (own) pool().release_stream(st, device, false); pool().release(d_in); } } back{ st, device, own_stream, nullptr };
	const u64 *d_src = in.d_words;
	i64 d_stride = in.d_stride;
	const uint32_t *d_dig = nullptr;
	const i64 *d_off = nullptr;
	// inputs up to 32 KiB: no copy at all -- the kernel reads the pinned staging buffer over the link (two batched round trips;
	// 2-6 us less than a host-to-device copy in front of the launch: profiles/r04_small.txt).  GF2BV_SMALL_ZC=0 / 1 forces either.
	bool zero_copy = in_bytes <= ((size_t)32 << 10);
	if (const char *e = getenv("GF2BV_SMALL_ZC")) zero_copy = atoi(e) != 0;
	if (in_bytes) {
		if (!zero_copy) HIPCHK(pool().alloc(&back.d_in, in_bytes, device));
		char *d_base = zero_copy ? G.h_in : (char *)back.d_in;
		if (in.h_digits) {
			memcpy(G.h_in, in.h_off, sizeof(i64) * (size_t)(rows + 1));
			if (ndig) memcpy(G.h_in + sizeof(i64) * (size_t)(rows + 1), in.h_digits, sizeof(uint32_t) * (size_t)ndig);
			d_off = (const i64 *)d_base;
			d_dig = (const uint32_t *)(d_base + sizeof(i64) * (size_t)(rows + 1));
		} else {
			for (i64 r = 0; r < rows; r++) memcpy(G.h_in + sizeof(u64) * (size_t)(r * wt), in.h_words + r * in.h_stride, sizeof(u64) * (size_t)wt);
			d_src = (const u64 *)d_base; d_stride = wt;
		}
		if (!zero_copy) HIPCHK(hipMemcpyAsync(back.d_in, G.h_in, in_bytes, hipMemcpyHostToDevice, st));
	}
	// 16 wavefronts: every phase but wavefront 0's column loop is a few dependent LDS round trips per item, i.e. latency --
	// 640 x 256: 143 us with 256 threads (profiles/r04_small.txt)
	constexpr int NT = 1024;
	const size_t lds = sizeof(u64) * (size_t)((rows + 512) * small_pitch(wt)) + sizeof(SmallLds);
	{
		static std::mutex mu;
		static std::map<int, bool> raised;          // device -> the > 64 KiB dynamic-LDS attribute has been raised there
		std::lock_guard<std::mutex> lk(mu);
		if (!raised[device]) {
			HIPCHK(hipFuncSetAttribute((const void *)k_small_solve<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
			raised[device] = true;
		}
	}
	memset(G.h_out, 0, 16);
	k_small_solve<NT><<<dim3(1), dim3(NT), lds, st>>>(d_src, d_stride, d_dig, d_off, in.bpd, (int)rows, (int)cols, want_basis ? 1 : 0, (unsigned *)G.h_out,
	                                                   probe ? (unsigned long long *)(G.h_out + res_bytes) : nullptr);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(st));
	const unsigned *ho = (const unsigned *)G.h_out;
	if (probe) {
		const unsigned long long *p = (const unsigned long long *)(G.h_out + res_bytes);
		fprintf(stderr, "[gf2bv small %lld x %lld] us: load %.1f candidates %.1f wavefront0 %.1f candidate rows %.1f tables %.1f all rows %.1f read-out %.1f; passes %llu\n",
		        (long long)rows, (long long)cols, p[0] / 100.0, p[1] / 100.0, p[2] / 100.0, p[3] / 100.0, p[4] / 100.0, p[5] / 100.0, p[6] / 100.0, p[7]);
	}
	const i64 rank = ho[0];
	const bool bad = ho[1] != 0;
	if (rank < 0 || rank > cols) return fail(GF2BV_ERR_HIP, "small solve returned an impossible rank");
	gf2bv_result *R = new gf2bv_result();
	R->status = bad ? GF2BV_STATUS_INCONSISTENT : GF2BV_STATUS_SOLVED;
	R->rank = rank; R->cw = cw; R->dim = cols - rank;
	const int32_t *pv = (const int32_t *)(G.h_out + 16);
	R->pivots.assign(pv, pv + rank);
	R->origin.assign(std::max<i64>(1, cw), 0);
	const u64 *org = (const u64 *)(G.h_out + head);
	if (!bad) {
		std::copy(org, org + cw, R->origin.begin());
		if (want_basis) {
			// free columns in M4RI's kernel order (_internal.c:348); the kernel wrote the vectors in ascending column order
			std::vector<int> order(cols), slot(cols, -1);
			for (i64 i = 0; i < cols; i++) order[i] = (int)i;
			for (i64 i = 0; i < rank; i++) std::swap(order[i], order[pv[i]]);
			std::vector<char> isp(cols, 0);
			for (i64 i = 0; i < rank; i++) isp[pv[i]] = 1;
			int j = 0;
			for (i64 c = 0; c < cols; c++) if (!isp[c]) slot[c] = j++;
			const u64 *Y = org + cw;
			R->basis.assign((size_t)R->dim * std::max<i64>(1, cw), 0);
			for (i64 t = 0; t < R->dim; t++) {
				const int f = order[rank + t];
				u64 *v = R->basis.data() + (size_t)t * cw;
				std::copy(Y + (size_t)slot[f] * cw, Y + (size_t)(slot[f] + 1) * cw, v);
				v[f >> 6] |= 1ull << (f & 63);
			}
		}
	}
	gf2bv_stats &s = R->stats;
	s.rows = rows; s.cols = cols; s.stride_words = wt;
	s.rank = rank; s.dimension = R->dim; s.status = R->status;
	s.n_panels = (int)cw; s.gang_systems = 1; s.tile_words = 0;
	s.small_path = 1;
	s.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
	*out = R;
	return GF2BV_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int gf2bv_version(void) { return 100; }

// sha256 (first 16 hex digits) over gf2_solver.hip + gf2_kernels.hip.h + gf2bv_hip.h, handed in by gf2bv_amd/build.py; the
// marker in front lets build.py read the id of an existing library without loading it (staleness by content, not by mtime)
#ifndef GF2BV_BUILD_ID
#define GF2BV_BUILD_ID "unknown"
#endif
static const char g_build_id[] = "GF2BV_BUILD_ID=" GF2BV_BUILD_ID;
const char *gf2bv_build_id(void) { return g_build_id + 15; }

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
	return guarded([&]() -> int {
	if (!out || !d_aug) return fail(GF2BV_ERR_ARG, "null pointer");
	*out = nullptr;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	if (stride_words % 2 != 0 || stride_words < (cols + 1 + 63) / 64 || ((uintptr_t)d_aug & 15))
		return fail(GF2BV_ERR_ARG, "device matrix needs 16-byte alignment and an even stride_words covering cols+1 bits");
	rc = check_device(device);
	if (rc) return rc;
	if (small_eligible(rows, cols) && !time_kernels) {
		SmallInput in; in.d_words = (const u64 *)d_aug; in.d_stride = stride_words;
		return small_solve(in, rows, cols, mode, device, (hipStream_t)stream, out);
	}
	Solver S;
	S.t_begin = std::chrono::steady_clock::now();
	S.device = device;
	S.sA = (hipStream_t)stream;
	S.src = (const u64 *)d_aug;
	S.rows = rows; S.cols = cols; S.stride = stride_words; S.mode = mode;
	S.time_kernels = time_kernels != 0;
	rc = solver_enqueue(S);
	if (rc) return rc;
	return solver_finish(S, out);
	});
}

int gf2bv_solve_batch_device(void *d_aug, int64_t nsys, int64_t sys_stride_words, int64_t rows, int64_t cols,
                             int64_t stride_words, int mode, int device, void *stream, int time_kernels, gf2bv_result **out)
{
	return guarded([&]() -> int {
	if (!out || !d_aug || nsys < 0) return fail(GF2BV_ERR_ARG, "null pointer");
	for (i64 s = 0; s < nsys; s++) out[s] = nullptr;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	if (stride_words % 2 != 0 || stride_words < (cols + 1 + 63) / 64 || sys_stride_words < rows * stride_words ||
	    ((uintptr_t)d_aug & 15) || (sys_stride_words & 1))
		return fail(GF2BV_ERR_ARG, "bad batch layout");
	rc = check_device(device);
	if (rc) return rc;
	// Independent systems of one shape run as GANGS: the forward elimination of `gang` systems is one set
	// of launches (blockIdx.y = system), so the latency-bound panel path and the host's ~3500 launches
	// per elimination are shared by the whole gang and the bulk updates of all its systems fill the chip.
	// NS host threads each take every NS-th gang on their own stream pair (one gang's back-substitution
	// and export then overlap the next gang's elimination).
	const i64 gang = pick_gang(nsys, rows, cols);
	// Gangs in flight: NS host threads, each with its own stream pair, take the next gang off a shared list.  Two threads that
	// start together stay in phase for the whole job -- both gangs bulk-bound at once, both in their tails at once.  Starting them
	// apart (the list beginning with part gangs, thread t's first gang (t + 1) / NS of a full one) was built and
	// measured in round 4: 3.56 ms per system against 3.42 in phase on 192 x 32768^2 (profiles/r04_batch_scans.txt) -- the bulk
	// update runs throughout a gang's elimination (its launches are in flight 89 % of the wall time), there is no idle tail to
	// fill, and part gangs only make smaller launches.  Not the default.
	std::vector<std::pair<i64, int>> ranges;       // (first system, systems)
	int NS = 2;
	if (const char *e = getenv("GF2BV_BATCH_THREADS")) { int v = atoi(e); if (v >= 1) NS = std::min(v, 16); }
	{
		i64 s0 = 0;
		while (s0 < nsys) { const i64 ns = std::min<i64>(gang, nsys - s0); ranges.emplace_back(s0, (int)ns); s0 += ns; }
	}
	const i64 ngangs = (i64)ranges.size();
	NS = (int)std::min<i64>(ngangs, NS);
	std::atomic<i64> next_gang{0};
	// Ordering contract: the matrices are whatever `stream` (the caller's stream, NULL = the null stream) has
	// produced when this call is made.  The gangs run on the library's own non-blocking streams, which are not
	// ordered against any other stream by themselves: each waits for an event recorded on `stream` here.
	struct Ready {
		hipEvent_t ev = nullptr;
		~Ready() { pool().release_event(ev, false); }
	} ready;
	HIPCHK(pool().event(&ready.ev, false));
	HIPCHK(hipEventRecord(ready.ev, (hipStream_t)stream));
	std::vector<int> rcs(NS, GF2BV_OK);
	std::vector<std::string> errs(NS);
	HIPCHK(hipSetDevice(device));
	std::vector<std::thread> workers;
	struct Join { std::vector<std::thread> &w; ~Join() { for (auto &t : w) if (t.joinable()) t.join(); } } joiner{ workers };
	for (int t = 0; t < NS; t++) {
		workers.emplace_back([&, t]() {
			if (hipSetDevice(device) != hipSuccess) { rcs[t] = GF2BV_ERR_HIP; errs[t] = "hipSetDevice"; return; }
			hipStream_t st = nullptr;
			const bool own_st = true;
			if (pool().stream(&st, device, 2) != hipSuccess) { rcs[t] = GF2BV_ERR_HIP; errs[t] = "hipStreamCreate"; return; }
			if (hipStreamWaitEvent(st, ready.ev, 0) != hipSuccess) { rcs[t] = GF2BV_ERR_HIP; errs[t] = "hipStreamWaitEvent"; }
			for (i64 q; (q = next_gang.fetch_add(1)) < ngangs && rcs[t] == GF2BV_OK;) {
				try {
				const i64 s0 = ranges[(size_t)q].first;
				const int ns = ranges[(size_t)q].second;
				int rc = GF2BV_OK;
				// an expired hand-over gate voids THIS gang only: its results (if any were built) are dropped and the
				// gang runs once more with events -- the device is marked by then; other gangs' results stay
				for (int attempt = 0; attempt < 2; attempt++) {
					g_attempt = attempt;                        // (this worker thread's: gf2bv_stats::handover_retries of the gang)
					Solver S;
					S.t_begin = std::chrono::steady_clock::now();
					S.device = device;
					S.sA = st;
					S.nsys = ns;
					S.src = (const u64 *)d_aug + s0 * sys_stride_words;
					S.src_sys_words = sys_stride_words;
					S.rows = rows; S.cols = cols; S.stride = stride_words; S.mode = mode;
					S.time_kernels = time_kernels != 0;
					rc = solve_gang(S, &out[s0]);
					if (rc != GF2BV_RETRY_EVENTS) break;
					for (int k = 0; k < ns; k++) { delete out[s0 + k]; out[s0 + k] = nullptr; }
					(void)hipStreamSynchronize(st);
				}
				if (rc == GF2BV_RETRY_EVENTS) { rc = GF2BV_ERR_HIP; g_err = "a stream hand-over gate timed out on the device"; }
				if (rc != GF2BV_OK) { rcs[t] = rc; errs[t] = g_err; }
				} catch (const std::bad_alloc &) { rcs[t] = GF2BV_ERR_NOMEM; errs[t] = "out of host memory"; }
				(void)hipStreamSynchronize(st);
			}
			if (own_st) pool().release_stream(st, device, 2);
		});
	}
	for (auto &w : workers) w.join();
	for (int t = 0; t < NS; t++)
		if (rcs[t] != GF2BV_OK) return fail(rcs[t], errs[t].c_str());
	return GF2BV_OK;
	});
}

int gf2bv_solve_words(const uint64_t *aug, int64_t rows, int64_t cols, int64_t stride_words, int mode,
                      int device, gf2bv_result **out)
{
	return guarded([&]() -> int {
	if (!out || (!aug && rows > 0)) return fail(GF2BV_ERR_ARG, "null pointer");
	*out = nullptr;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	const i64 wt = (cols + 1 + 63) / 64;
	if (stride_words < wt) return fail(GF2BV_ERR_ARG, "stride_words does not cover cols+1 bits");
	rc = check_device(device);
	if (rc) return rc;
	if (small_eligible(rows, cols)) {
		SmallInput in; in.h_words = reinterpret_cast<const u64 *>(aug); in.h_stride = stride_words;
		return small_solve(in, rows, cols, mode, device, nullptr, out);
	}
	Solver S;
	S.t_begin = std::chrono::steady_clock::now();
	S.device = device;
	HIPCHK(pool().stream(&S.sA, device, false));
	S.own_sA = true;
	S.rows = rows; S.cols = cols; S.mode = mode;
	S.stride = wt;
	HIPCHK(pool().alloc((void **)&S.tmp_src, sizeof(u64) * std::max<i64>(1, rows) * S.stride, device));
	S.src = S.tmp_src;
	Scratch scratch;                  // (declared after S: released first, after synchronising the solve's stream)
	scratch.sync_first = S.sA;
	hipEvent_t p0, p1;
	HIPCHK(scratch.event(&p0)); HIPCHK(scratch.event(&p1));
	HIPCHK(hipEventRecord(p0, S.sA));
	if (rows > 0)
		HIPCHK(hipMemcpy2DAsync(S.tmp_src, S.stride * 8, aug, stride_words * 8, wt * 8, rows, hipMemcpyHostToDevice, S.sA));
	// bits above column `cols` are ignored by the reference (_internal.c:414): they are never
	// pivot candidates (colmask), never exported, and the RHS is read at exactly column `cols`.
	HIPCHK(hipEventRecord(p1, S.sA));
	rc = solver_enqueue(S);
	if (rc == GF2BV_OK) {
		(void)hipEventSynchronize(p1);
		(void)hipEventElapsedTime(&S.ms_pack, p0, p1);
		rc = solver_finish(S, out);
	}
	return rc;
	});
}

// One device's share of a digits batch.  digit_off[] holds ABSOLUTE digit positions (a later share of a larger batch does
// not start at 0): only digits[digit_off[0] .. digit_off[nsys * rows]) are uploaded and the pack kernel sees them through a
// rebased pointer.
static int batch_digits_on(const uint32_t *digits, const int64_t *digit_off, int bits_per_digit, int64_t nsys,
                           int64_t rows, int64_t cols, int mode, int device, gf2bv_result **out)
{
	bool again = false;
	return guarded([&]() -> int {
	if (!out || !digit_off || nsys < 0) return fail(GF2BV_ERR_ARG, "null pointer");
	for (i64 s = 0; s < nsys; s++) { if (again) delete out[s]; out[s] = nullptr; }      // (a retry drops what the voided attempt built)
	again = true;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	if (bits_per_digit < 1 || bits_per_digit > 32) return fail(GF2BV_ERR_ARG, "bits_per_digit must be 1..32");
	rc = check_device(device);
	if (rc) return rc;
	if (nsys == 0) return GF2BV_OK;
	HIPCHK(hipSetDevice(device));
	const i64 wt = (cols + 1 + 63) / 64, ntiles = tiles_for(wt), srows = slab_rows(rows);
	const i64 m_stride = ntiles * TW * srows;
	const i64 nrows_all = nsys * rows, dig0 = digit_off[0], ndig = digit_off[nrows_all] - dig0;
	if (ndig < 0) return fail(GF2BV_ERR_ARG, "digit offsets must not decrease");
	for (i64 r = 0; r < nrows_all; r++)          // (the pack kernel reads digits[off[r] .. off[r + 1]) of the uploaded share)
		if (digit_off[r + 1] < digit_off[r]) return fail(GF2BV_ERR_ARG, "digit offsets must not decrease");
	// The offsets go up once; the DIGITS go up gang by gang, each on the stream of the host thread that takes the gang (round 5: two
	// threads, as in gf2bv_solve_batch_device -- one gang's upload and pack run under the other's elimination, and two latency-bound
	// gangs of sparse systems overlap; before: one upload of everything, then the gangs one after the other on one stream -- 16 MT19937
	// recovery systems 43 ms, 15 of them the upload).  Every gang packs its own systems straight into tile-major slabs.
	struct Staged {
		i64 *off = nullptr; hipStream_t st = nullptr; hipEvent_t ready = nullptr; int device = 0;
		~Staged()
		{
			if (st) (void)hipStreamSynchronize(st);
			pool().release(off);
			pool().release_event(ready, false);
			if (st) pool().release_stream(st, device, 2);
		}
	} G;
	G.device = device;
	HIPCHK(pool().stream(&G.st, device, 2));
	HIPCHK(pool().alloc((void **)&G.off, sizeof(i64) * (nrows_all + 1), device));
	HIPCHK(hipMemcpyAsync(G.off, digit_off, sizeof(i64) * (nrows_all + 1), hipMemcpyHostToDevice, G.st));
	HIPCHK(pool().event(&G.ready, false));
	HIPCHK(hipEventRecord(G.ready, G.st));
	const i64 gang = pick_gang(nsys, rows, cols);
	const i64 ngangs = (nsys + gang - 1) / gang;
	i64 max_dig = 1;
	for (i64 q = 0; q < ngangs; q++)
		max_dig = std::max<i64>(max_dig, digit_off[std::min<i64>(nsys, (q + 1) * gang) * rows] - digit_off[q * gang * rows]);
	int NS = 2;
	if (const char *e = getenv("GF2BV_BATCH_THREADS")) { int v = atoi(e); if (v >= 1) NS = std::min(v, 16); }
	NS = (int)std::min<i64>(ngangs, NS);
	std::atomic<i64> next_gang{0};
	std::vector<int> rcs((size_t)NS, GF2BV_OK);
	std::vector<std::string> errs((size_t)NS);
	const int attempt0 = g_attempt;                    // (a whole-call retry by guarded() reaches the workers' solvers)
	auto worker = [&](int t) {
		struct Mine {
			hipStream_t st = nullptr; uint32_t *dig = nullptr; int device = 0; bool own = true;
			~Mine()
			{
				if (st) (void)hipStreamSynchronize(st);
				pool().release(dig);
				if (st && own) pool().release_stream(st, device, 2);
			}
		} W;
		W.device = device;
		auto run = [&]() -> int {
			HIPCHK(hipSetDevice(device));
			HIPCHK(pool().stream(&W.st, device, 2));
			HIPCHK(hipStreamWaitEvent(W.st, G.ready, 0));
			HIPCHK(pool().alloc((void **)&W.dig, sizeof(uint32_t) * max_dig, device));
			for (i64 q; (q = next_gang.fetch_add(1)) < ngangs;) {
				const i64 s0 = q * gang;
				const int ns = (int)std::min<i64>(gang, nsys - s0);
				const i64 d0 = digit_off[s0 * rows], nd = digit_off[(s0 + ns) * rows] - d0;
				if (nd) HIPCHK(hipMemcpyAsync(W.dig, digits + d0, sizeof(uint32_t) * nd, hipMemcpyHostToDevice, W.st));
				int rc = GF2BV_OK;
				// an expired hand-over gate voids THIS gang only (as in gf2bv_solve_batch_device): packed and solved once more with events
				for (int attempt = attempt0; attempt < 2; attempt++) {
					g_attempt = attempt;
					Solver S;
					S.t_begin = std::chrono::steady_clock::now();
					S.device = device;
					S.sA = W.st;
					S.nsys = ns;
					S.rows = rows; S.cols = cols; S.mode = mode;
					S.stride = ntiles * TW;
					HIPCHK(pool().alloc((void **)&S.M, sizeof(u64) * m_stride * S.nsys + kOuterSlackBytes, device));
					if (rows * ntiles * TW > 0)
						k_pack_digits<<<dim3((unsigned)((ntiles * TW + 255) / 256), (unsigned)std::min<i64>(rows, 65535), S.nsys), dim3(256), 0, S.sA>>>(
							W.dig, G.off + s0 * rows, bits_per_digit, (i64)rows, (i64)cols, ntiles * TW, srows, S.M, SysStride{m_stride, 0}, d0);
					HIPCHK(hipGetLastError());
					rc = solve_gang(S, &out[s0]);
					if (rc != GF2BV_RETRY_EVENTS) break;
					for (int k = 0; k < ns; k++) { delete out[s0 + k]; out[s0 + k] = nullptr; }
					(void)hipStreamSynchronize(W.st);
				}
				if (rc == GF2BV_RETRY_EVENTS) return fail(GF2BV_ERR_HIP, "a stream hand-over gate timed out on the device");
				if (rc != GF2BV_OK) return rc;
				HIPCHK(hipStreamSynchronize(W.st));          // (the next gang's digits overwrite W.dig)
			}
			return GF2BV_OK;
		};
		try { rcs[(size_t)t] = run(); }
		catch (const std::bad_alloc &) { rcs[(size_t)t] = fail(GF2BV_ERR_NOMEM, "out of host memory"); }
		catch (const std::exception &e) { rcs[(size_t)t] = fail(GF2BV_ERR_HIP, e.what()); }
		if (rcs[(size_t)t] != GF2BV_OK) { errs[(size_t)t] = g_err; next_gang.store(ngangs); }
	};
	{
		std::vector<std::thread> workers;
		struct Join { std::vector<std::thread> &w; ~Join() { for (auto &t : w) if (t.joinable()) t.join(); } } joiner{ workers };
		for (int t = 1; t < NS; t++) workers.emplace_back(worker, t);
		worker(0);
	}
	g_attempt = attempt0;
	for (int t = 0; t < NS; t++)
		if (rcs[(size_t)t] != GF2BV_OK) return fail(rcs[(size_t)t], errs[(size_t)t].c_str());
	return GF2BV_OK;
	});
}

int gf2bv_solve_batch_digits(const uint32_t *digits, const int64_t *digit_off, int bits_per_digit, int64_t nsys,
                             int64_t rows, int64_t cols, int mode, int device, gf2bv_result **out)
{
	return batch_digits_on(digits, digit_off, bits_per_digit, nsys, rows, cols, mode, device, out);
}

// The same batch over SEVERAL devices (the reference solves one system per m4ri_solve call on one core; independent
// systems -- one per output bit / per instance in the recovery examples -- are this path's natural shard unit):
// entry k of devices[] takes the k-th contiguous share of the systems on its own host thread (a device may be listed more
// than once: its shares then run as concurrent gangs on it), results land in out[] in input order.  No collective and no
// torch: one process drives every GPU of the node.
int gf2bv_solve_batch_digits_multi(const uint32_t *digits, const int64_t *digit_off, int bits_per_digit, int64_t nsys,
                                   int64_t rows, int64_t cols, int mode, const int *devices, int ndevices,
                                   gf2bv_result **out)
{
	if (!out || !digit_off || !devices || nsys < 0 || ndevices < 1) return fail(GF2BV_ERR_ARG, "null pointer");
	for (i64 s = 0; s < nsys; s++) out[s] = nullptr;
	for (int k = 0; k < ndevices; k++) { int rc = check_device(devices[k]); if (rc) return rc; }
	const int nshares = (int)std::min<i64>(ndevices, std::max<i64>(nsys, 1));
	if (nshares == 1) return batch_digits_on(digits, digit_off, bits_per_digit, nsys, rows, cols, mode, devices[0], out);
	std::vector<int> rcs((size_t)nshares, GF2BV_OK);
	std::vector<std::string> errs((size_t)nshares);
	std::vector<std::thread> th;
	try {
		for (int k = 0; k < nshares; k++) {
			const i64 lo = nsys * k / nshares, hi = nsys * (k + 1) / nshares;
			th.emplace_back([&, k, lo, hi]() {
				rcs[k] = batch_digits_on(digits, digit_off + lo * rows, bits_per_digit, hi - lo, rows, cols, mode, devices[k], out + lo);
				if (rcs[k] != GF2BV_OK) errs[k] = g_err;
			});
		}
	} catch (const std::exception &) { for (auto &t : th) t.join(); for (i64 s = 0; s < nsys; s++) { delete out[s]; out[s] = nullptr; } return fail(GF2BV_ERR_NOMEM, "could not start a host thread per device"); }
	for (auto &t : th) t.join();
	for (int k = 0; k < nshares; k++)
		if (rcs[k] != GF2BV_OK) {
			for (i64 s = 0; s < nsys; s++) { delete out[s]; out[s] = nullptr; }
			return fail(rcs[k], errs[k].c_str());
		}
	return GF2BV_OK;
}

int gf2bv_solve_digits(const uint32_t *digits, const int64_t *digit_off, int bits_per_digit, int64_t rows,
                       int64_t cols, int mode, int device, gf2bv_result **out)
{
	return guarded([&]() -> int {
	if (!out || !digit_off) return fail(GF2BV_ERR_ARG, "null pointer");
	*out = nullptr;
	int rc = check_shape(rows, cols, mode);
	if (rc) return rc;
	if (bits_per_digit < 1 || bits_per_digit > 32) return fail(GF2BV_ERR_ARG, "bits_per_digit must be 1..32");
	rc = check_device(device);
	if (rc) return rc;
	if (digit_off[0] != 0) return fail(GF2BV_ERR_ARG, "digit offsets must start at 0");
	for (i64 r = 0; r < rows; r++)
		if (digit_off[r + 1] < digit_off[r]) return fail(GF2BV_ERR_ARG, "digit offsets must not decrease");
	if (small_eligible(rows, cols)) {
		SmallInput in; in.h_digits = digits; in.h_off = reinterpret_cast<const i64 *>(digit_off); in.bpd = bits_per_digit;
		return small_solve(in, rows, cols, mode, device, nullptr, out);
	}
	Solver S;
	S.t_begin = std::chrono::steady_clock::now();
	S.device = device;
	HIPCHK(pool().stream(&S.sA, device, false));
	S.own_sA = true;
	S.rows = rows; S.cols = cols; S.mode = mode;
	const i64 wt = (cols + 1 + 63) / 64;
	const i64 ntiles = tiles_for(wt);
	S.stride = ntiles * TW;
	HIPCHK(pool().alloc((void **)&S.M, sizeof(u64) * ntiles * TW * slab_rows(rows) + kOuterSlackBytes, device));     // packed straight into tiles
	const i64 ndig = digit_off[rows];
	Scratch scratch;                  // digits, offsets and the pack events go back to the pool on every path
	scratch.sync_first = S.sA;
	uint32_t *d_dig = nullptr;
	i64 *d_off = nullptr;
	HIPCHK(scratch.alloc((void **)&d_dig, sizeof(uint32_t) * std::max<i64>(1, ndig), device));
	HIPCHK(scratch.alloc((void **)&d_off, sizeof(i64) * (rows + 1), device));
	hipEvent_t p0, p1;
	HIPCHK(scratch.event(&p0)); HIPCHK(scratch.event(&p1));
	HIPCHK(hipEventRecord(p0, S.sA));
	if (ndig) HIPCHK(hipMemcpyAsync(d_dig, digits, sizeof(uint32_t) * ndig, hipMemcpyHostToDevice, S.sA));
	HIPCHK(hipMemcpyAsync(d_off, digit_off, sizeof(i64) * (rows + 1), hipMemcpyHostToDevice, S.sA));
	{
		i64 total = rows * ntiles * TW;
		if (total > 0)
			k_pack_digits<<<dim3((unsigned)((ntiles * TW + 255) / 256), (unsigned)std::min<i64>(rows, 65535)), dim3(256), 0, S.sA>>>(d_dig, d_off, bits_per_digit, (i64)rows,
			                                                                            (i64)cols, ntiles * TW, slab_rows(rows), S.M,
			                                                                            SysStride{0, 0}, (i64)0);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(p1, S.sA));
	rc = solver_enqueue(S);
	if (rc == GF2BV_OK) {
		(void)hipEventSynchronize(p1);
		(void)hipEventElapsedTime(&S.ms_pack, p0, p1);
		rc = solver_finish(S, out);
	}
	return rc;
	});
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

// ---- column-slab solve: ONE system over several GPUs ---------------------------------
// Rank r of `world` owns the column tiles t with t % world == r (cyclic: the trailing matrix shrinks from the left).
// Per block of G panels:
//   * the rank that owns the tile of the block's window runs the panel path there (gf2bv_slab_factor) -- exactly
//     the single-GPU kernels on its compact window, the window of the NEXT block it owns being carried forward by
//     k_prio_window as before -- and exports the block's records: SolveState, alive bound, PanelRec / PanelAux of its
//     panels and the per-row multipliers (G x rows x 8 bytes): the payload of ONE broadcast over xGMI;
//   * every rank imports the payload (gf2bv_slab_apply) and runs k_block_trsm + k_update on the tiles it owns.
// After the last block each rank moves its pivot rows' parked window words into its tiles (gf2bv_slab_finish_local);
// the ranks' tiles are then gathered on one rank (the caller's collective: the working matrix is the caller's
// buffer), which has every record and finishes like a single-GPU solve (gf2bv_slab_solve: consistency check,
// back-substitution, export).  The communication is the caller's (torch.distributed: RCCL on GPUs, gloo in the CPU-side
// tests): this library stays free of it, like the rest of the C ABI.
struct gf2bv_slab {
	Solver S;
	size_t payload_bytes = 0;
	size_t o_blk = 0, o_pan = 0, o_aux = 0, o_mult = 0;
	int next_prio = 0;         // k_prio_window(b) has been enqueued for all b < next_prio (on this rank, where it owns the next window)
	// stream-ordered factor / apply (gf2bv_slab_factor_on / _apply_on): hand-overs between the caller's collective stream and
	// the library's panel stream
	// the panel stream, per payload buffer (block parity): [records exported, collective enqueued, records imported]
	hipEvent_t ev_factored[2] = { nullptr, nullptr }, ev_bcast[2] = { nullptr, nullptr }, ev_imported[2] = { nullptr, nullptr };
	bool bcast_seen[2] = { false, false }, import_seen[2] = { false, false };
	~gf2bv_slab()
	{
		if (S.sA) (void)hipStreamSynchronize(S.sA);       // (nothing that waits on these events may still be queued)
		if (S.sB) (void)hipStreamSynchronize(S.sB);
		for (int k = 0; k < 2; k++) for (hipEvent_t e : { ev_factored[k], ev_bcast[k], ev_imported[k] }) pool().release_event(e, false);
	}
};

namespace {
int slab_owner_of_block(const Solver &S, int b) { return ((b * S.impl->G) >> GF2_OWN_LOG) % S.world; }
}

int gf2bv_slab_open(void *d_aug, int64_t rows, int64_t cols, int64_t stride_words, void *d_work, int64_t work_words,
                    int world, int rank, int device, gf2bv_slab **out)
{
	return guarded([&]() -> int {
	if (!out || !d_aug || !d_work || world < 1 || rank < 0 || rank >= world) return fail(GF2BV_ERR_ARG, "bad slab arguments");
	*out = nullptr;
	int rc = check_shape(rows, cols, GF2BV_MODE_SINGLE);
	if (rc) return rc;
	if (stride_words % 2 != 0 || stride_words < (cols + 1 + 63) / 64 || ((uintptr_t)d_aug & 15) || ((uintptr_t)d_work & 15))
		return fail(GF2BV_ERR_ARG, "device matrix needs 16-byte alignment and an even stride_words covering cols+1 bits");
	if (work_words < gf2bv_slab_work_words(rows, cols)) return fail(GF2BV_ERR_ARG, "working matrix too small (gf2bv_slab_work_words)");
	rc = check_device(device);
	if (rc) return rc;
	gf2bv_slab *h = new gf2bv_slab();
	Solver &S = h->S;
	S.t_begin = std::chrono::steady_clock::now();
	S.device = device;
	hipError_t e = pool().stream(&S.sA, device, false);
	if (e != hipSuccess) { delete h; return fail(GF2BV_ERR_HIP, "hipStreamCreate", e); }
	S.own_sA = true;
	S.src = (const u64 *)d_aug;
	S.rows = rows; S.cols = cols; S.stride = stride_words; S.mode = GF2BV_MODE_SINGLE;
	S.world = world; S.wrank = rank;
	S.M = (u64 *)d_work; S.ext_M = true;
	rc = solver_alloc(S);                           // (converts the whole input: tiles of other ranks are simply never touched again)
	if (rc == GF2BV_OK) rc = enqueue_forward_begin(S, slab_owner_of_block(S, 0) == rank);
	if (rc != GF2BV_OK) { delete h; return rc; }
	const i64 R = std::max<i64>(1, S.rows);
	const int G = S.impl->G;
	size_t off = 0;
	auto carve = [&](size_t bytes) { size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
	(void)carve(sizeof(SolveState)); h->o_blk = carve(sizeof(int)); h->o_pan = carve(sizeof(PanelRec) * G);
	h->o_aux = carve(sizeof(PanelAux) * G); h->o_mult = carve(sizeof(u64) * G * mult_rows(R));
	h->payload_bytes = off;
	for (int k = 0; k < 2; k++)
		for (hipEvent_t *e : { &h->ev_factored[k], &h->ev_bcast[k], &h->ev_imported[k] })
			if (pool().event(e, false) != hipSuccess) { delete h; return fail(GF2BV_ERR_HIP, "hipEventCreate"); }
	*out = h;
	return GF2BV_OK;
	});
}

int64_t gf2bv_slab_work_words(int64_t rows, int64_t cols)
{
	const i64 wt = (cols + 1 + 63) / 64, ntiles = tiles_for(wt);
	return ntiles * TW * slab_rows(rows);
}
int64_t gf2bv_slab_tiles(int64_t cols) { return tiles_for((cols + 1 + 63) / 64) * TW >> GF2_OWN_LOG; }      // ownership units of 8 words
int64_t gf2bv_slab_blocks(const gf2bv_slab *h) { return h ? h->S.nblocks : -1; }
int gf2bv_slab_owner(const gf2bv_slab *h, int block) { return h ? slab_owner_of_block(h->S, block) : -1; }
int64_t gf2bv_slab_payload_bytes(const gf2bv_slab *h) { return h ? (int64_t)h->payload_bytes : -1; }

// records of block b <-> payload, one launch on `st` (export: to_payload)
static int slab_copy_records(gf2bv_slab *h, int b, char *P, bool to_payload, hipStream_t st)
{
	Solver &S = h->S;
	const BlockGeom g = block_geom(S, b);
	const int G = S.impl->G;
	static_assert(sizeof(SolveState) % 4 == 0 && sizeof(PanelRec) % 4 == 0 && sizeof(PanelAux) % 4 == 0, "4-byte copy units");
	struct { void *rec; size_t off, bytes; } seg[5] = {
		{ S.st, 0, sizeof(SolveState) }, { S.blk_first + b, h->o_blk, sizeof(int) }, { S.panels + g.j0, h->o_pan, sizeof(PanelRec) * g.gb },
		{ S.aux + g.j0, h->o_aux, sizeof(PanelAux) * g.gb }, { g.mset, h->o_mult, sizeof(u64) * G * mult_rows(S.rows) } };
	SlabSegs sg;
	size_t total = 0;
	for (int k = 0; k < 5; k++) {
		sg.s[k].src = to_payload ? seg[k].rec : (void *)(P + seg[k].off);
		sg.s[k].dst = to_payload ? (void *)(P + seg[k].off) : seg[k].rec;
		sg.s[k].bytes = (unsigned)seg[k].bytes;
		total += seg[k].bytes;
	}
	const unsigned wgs = (unsigned)std::min<size_t>(256, std::max<size_t>(1, total / 16 / 256 / 4));
	k_slab_copy<<<dim3(wgs), dim3(256), 0, st>>>(sg);
	HIPCHK(hipGetLastError());
	return GF2BV_OK;
}

// Owner of block b: (carry its window forward,) factorise it, export the records into d_payload.
// Synchronous form (host_sync): returns when the payload is complete.  Stream-ordered form: `stream` is the stream the
// caller's collective runs on; the export is ONE launch on the panel stream behind the panel path, `stream` is made to wait
// for it and the call returns at once -- a broadcast enqueued on `stream` afterwards sends finished records, and the panel
// stream never waits for the collective: the caller alternates TWO payload buffers (block b uses buffer b & 1), so that the
// export of block b + 1 does not have to wait for the broadcast of block b (only for that of block b - 1, long gone).
static int slab_factor_on(gf2bv_slab *h, int b, void *d_payload, hipStream_t stream, bool host_sync)
{
	if (!h || !d_payload || b < 0 || b >= h->S.nblocks) return fail(GF2BV_ERR_ARG, "bad block");
	Solver &S = h->S;
	if (slab_owner_of_block(S, b) != S.wrank) return fail(GF2BV_ERR_ARG, "this rank does not own the block's window");
	HIPCHK(hipSetDevice(S.device));
	int rc;
	// the window of block b: carried forward from block b-1's records (imported by gf2bv_slab_apply(b-1)) -- needs this
	// rank's bulk update of block b-2, nothing newer: the look-ahead of the single-GPU solve, per owner
	if (b > 0 && (rc = enqueue_block_prio(S, b - 1))) return rc;
	if ((rc = enqueue_block_panel(S, b))) return rc;
	if (S.world == 1 && !host_sync) return GF2BV_OK;       // nobody imports: the records stay where they are
	const int par = b & 1;
	// the buffer of this parity: last read by the broadcast of block b - 2 on the caller's stream (recorded in slab_apply_on)
	if (!host_sync && h->bcast_seen[par]) HIPCHK(hipStreamWaitEvent(S.sA, h->ev_bcast[par], 0));
	if ((rc = slab_copy_records(h, b, (char *)d_payload, true, S.sA))) return rc;
	if (host_sync) HIPCHK(hipStreamSynchronize(S.sA));
	else {
		HIPCHK(hipEventRecord(h->ev_factored[par], S.sA));
		HIPCHK(hipStreamWaitEvent(stream, h->ev_factored[par], 0));
	}
	return GF2BV_OK;
}

// Every rank: take block b's records (the other ranks import them from the broadcast payload, which `stream` has produced),
// then TRSM + bulk update of the tiles this rank owns.  Asynchronous.
static int slab_apply_on(gf2bv_slab *h, int b, const void *d_payload, hipStream_t stream, bool host_sync)
{
	if (!h || !d_payload || b < 0 || b >= h->S.nblocks) return fail(GF2BV_ERR_ARG, "bad block");
	Solver &S = h->S;
	HIPCHK(hipSetDevice(S.device));
	const BlockGeom g = block_geom(S, b);
	const int par = b & 1;
	if (!host_sync && S.world > 1) {               // "the collective of block b has read / written buffer b & 1"
		HIPCHK(hipEventRecord(h->ev_bcast[par], stream));
		h->bcast_seen[par] = true;
	}
	if (slab_owner_of_block(S, b) != S.wrank) {
		if (!host_sync) HIPCHK(hipStreamWaitEvent(S.sA, h->ev_bcast[par], 0));      // the payload is what `stream` holds now
		// the multiplier set of block b was last read by this rank's bulk update of block b - nsets
		if (b >= S.nsets) HIPCHK(hipStreamWaitEvent(S.sA, S.waitPrio[b - S.nsets], 0));
		int rc = slab_copy_records(h, b, (char *)const_cast<void *>(d_payload), false, S.sA);
		if (rc) return rc;
		k_import_marks<<<dim3(1), dim3(256), 0, S.sA>>>(g.j0, g.gb, S.panels, S.aux, S.died, S.pivcol, S.urow);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(S.evA[b], S.sA));
		// the payload buffer may be overwritten once the import has read it: after this returns (host_sync), or by what the
		// caller enqueues on `stream` behind the wait below -- placed one call LATER (the import of the previous block: the
		// buffer it read is the one the next-but-one collective writes), so that the collective stream stays a block ahead
		if (host_sync) HIPCHK(hipStreamSynchronize(S.sA));
		else { HIPCHK(hipEventRecord(h->ev_imported[par], S.sA)); h->import_seen[par] = true; }
	}
	if (!host_sync && h->import_seen[par ^ 1]) HIPCHK(hipStreamWaitEvent(stream, h->ev_imported[par ^ 1], 0));
	return enqueue_block_bulk(S, b);
}

int gf2bv_slab_factor(gf2bv_slab *h, int b, void *d_payload)
{
	return guarded([&]() -> int { return slab_factor_on(h, b, d_payload, nullptr, true); });
}
int gf2bv_slab_apply(gf2bv_slab *h, int b, const void *d_payload)
{
	return guarded([&]() -> int { return slab_apply_on(h, b, d_payload, nullptr, true); });
}
int gf2bv_slab_factor_on(gf2bv_slab *h, int b, void *d_payload, void *stream)
{
	return guarded([&]() -> int { return slab_factor_on(h, b, d_payload, (hipStream_t)stream, false); });
}
int gf2bv_slab_apply_on(gf2bv_slab *h, int b, const void *d_payload, void *stream)
{
	return guarded([&]() -> int { return slab_apply_on(h, b, d_payload, (hipStream_t)stream, false); });
}

// After the last block: this rank's tiles are final (pivot rows' parked window words moved in); synchronous.
int gf2bv_slab_finish_local(gf2bv_slab *h)
{
	return guarded([&]() -> int {
	if (!h) return fail(GF2BV_ERR_ARG, "null pointer");
	Solver &S = h->S;
	HIPCHK(hipSetDevice(S.device));
	int rc = enqueue_forward_join(S);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(S.sA));
	HIPCHK(hipStreamSynchronize(S.sB));
	return GF2BV_OK;
	});
}

// On the rank that holds ALL tiles (after the caller's gather): consistency check, back-substitution, export -- the tail
// of a single-GPU solve_one.
int gf2bv_slab_solve(gf2bv_slab *h, gf2bv_result **out)
{
	return guarded([&]() -> int {
	if (!h || !out) return fail(GF2BV_ERR_ARG, "null pointer");
	*out = nullptr;
	Solver &S = h->S;
	HIPCHK(hipSetDevice(S.device));
	int rc = enqueue_check_rhs(S);
	if (rc == GF2BV_OK) rc = enqueue_backward_single(S);
	if (rc) return rc;
	rc = solver_finish(S, out);
	// (the elimination state of a slab handle cannot be replayed from here: an expired gate is an error, not a retry)
	if (rc == GF2BV_RETRY_EVENTS) return fail(GF2BV_ERR_HIP, "a stream hand-over gate timed out on the device");
	return rc;
	});
}

void gf2bv_slab_close(gf2bv_slab *h)
{
	if (!h) return;
	(void)hipSetDevice(h->S.device);
	delete h;
}

// ---- AffineSpace on the device: bulk enumeration ------------------------------------------------
struct gf2bv_space {
	int device = 0;
	int64_t dim = 0, words = 0;
	u64 *d_origin = nullptr, *d_basis = nullptr;
	u64 *d_out = nullptr;
	u64 *h_out = nullptr;         // pinned staging: the device-to-host copy runs at link speed, and callers that pass no
	size_t out_words = 0;         // output pointer read the elements straight from it (gf2bv_space_buffer)
	hipStream_t st = nullptr;
};

int gf2bv_space_open(const uint64_t *origin, const uint64_t *basis, int64_t dimension, int64_t words, int device,
                     gf2bv_space **out)
{
	return guarded([&]() -> int {
	if (!out || !origin || dimension < 0 || words <= 0 || (dimension > 0 && !basis)) return fail(GF2BV_ERR_ARG, "bad space");
	*out = nullptr;
	int rc = check_device(device);
	if (rc) return rc;
	gf2bv_space *sp = new gf2bv_space();
	sp->device = device; sp->dim = dimension; sp->words = words;
	auto bail = [&](hipError_t e, const char *what) { gf2bv_space_close(sp); return fail(GF2BV_ERR_HIP, what, e); };
	hipError_t e = pool().stream(&sp->st, device, false);
	if (e != hipSuccess) return bail(e, "hipStreamCreate");
	if ((e = pool().alloc((void **)&sp->d_origin, sizeof(u64) * words, device)) != hipSuccess) return bail(e, "hipMalloc");
	if ((e = pool().alloc((void **)&sp->d_basis, sizeof(u64) * std::max<i64>(1, dimension * words), device)) != hipSuccess) return bail(e, "hipMalloc");
	if ((e = hipMemcpyAsync(sp->d_origin, origin, sizeof(u64) * words, hipMemcpyHostToDevice, sp->st)) != hipSuccess) return bail(e, "hipMemcpy");
	if (dimension > 0 && (e = hipMemcpyAsync(sp->d_basis, basis, sizeof(u64) * dimension * words, hipMemcpyHostToDevice, sp->st)) != hipSuccess)
		return bail(e, "hipMemcpy");
	if ((e = hipStreamSynchronize(sp->st)) != hipSuccess) return bail(e, "hipStreamSynchronize");      // the caller's buffers may go away
	*out = sp;
	return GF2BV_OK;
	});
}

int gf2bv_space_enumerate(gf2bv_space *sp, uint64_t first, int64_t count, int gray, uint64_t *out_words)
{
	return guarded([&]() -> int {
	if (!sp || count < 0) return fail(GF2BV_ERR_ARG, "null pointer");
	if (count == 0) return GF2BV_OK;
	HIPCHK(hipSetDevice(sp->device));
	const size_t need = (size_t)count * sp->words;
	if (need > sp->out_words) {
		pool().release(sp->d_out); sp->d_out = nullptr; sp->out_words = 0;
		if (sp->h_out) { (void)hipHostFree(sp->h_out); sp->h_out = nullptr; }
		HIPCHK(pool().alloc((void **)&sp->d_out, sizeof(u64) * need, sp->device));
		HIPCHK(hipHostMalloc((void **)&sp->h_out, sizeof(u64) * need, hipHostMallocDefault));
		sp->out_words = need;
	}
	const dim3 grid((unsigned)((sp->words + 255) / 256), (unsigned)std::min<i64>(count, 65535));
	k_space_enumerate<<<grid, dim3(256), 0, sp->st>>>(sp->d_origin, sp->d_basis, (int)std::min<i64>(sp->dim, 64), sp->words,
	                                                  (u64)first, (i64)count, gray, sp->d_out);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(sp->h_out, sp->d_out, sizeof(u64) * need, hipMemcpyDeviceToHost, sp->st));
	HIPCHK(hipStreamSynchronize(sp->st));
	if (out_words) memcpy(out_words, sp->h_out, sizeof(u64) * need);
	return GF2BV_OK;
	});
}

const uint64_t *gf2bv_space_buffer(const gf2bv_space *sp) { return sp ? reinterpret_cast<const uint64_t *>(sp->h_out) : nullptr; }

void gf2bv_space_close(gf2bv_space *sp)
{
	if (!sp) return;
	(void)hipSetDevice(sp->device);
	if (sp->st) (void)hipStreamSynchronize(sp->st);
	pool().release(sp->d_origin); pool().release(sp->d_basis); pool().release(sp->d_out);
	if (sp->h_out) (void)hipHostFree(sp->h_out);
	if (sp->st) pool().release_stream(sp->st, sp->device, false);
	delete sp;
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

// Registers (VGPRs per lane) and static LDS of the kernels that must fit on a CU TOGETHER: the default bulk-update instance
// and the panel-path kernels that run beside it.  out[2 k], out[2 k + 1] = registers, LDS bytes of kernel k (see
// gf2bv_hip.h for the order).
int gf2bv_kernel_resources(int device, int32_t *out, int n)
{
	return guarded([&]() -> int {
	if (!out || n < 10) return fail(GF2BV_ERR_ARG, "need room for 5 kernels");
	for (int k = 0; k < n; k++) out[k] = 0;
	int rc = check_device(device);
	if (rc) return rc;
	HIPCHK(hipSetDevice(device));
	const void *fn[5] = { (const void *)k_update16<512, false, 3, true, 512>, (const void *)k_block_fast, (const void *)k_narrow_all,
	                      (const void *)k_prio_window, (const void *)k_panel_step };
	for (int k = 0; k < 5; k++) {
		hipFuncAttributes a{};
		HIPCHK(hipFuncGetAttributes(&a, fn[k]));
		out[2 * k] = a.numRegs;
		out[2 * k + 1] = (int32_t)a.sharedSizeBytes;
	}
	if (n >= 13) {                 // the outer pass of the two-level elimination: registers, LDS, scratch bytes per lane (must be 0)
		hipFuncAttributes a{};
		HIPCHK(hipFuncGetAttributes(&a, (const void *)k_update16k_wide));       // (the default shape: 16 wavefronts under a budget of 120 registers)
		out[10] = a.numRegs; out[11] = (int32_t)a.sharedSizeBytes; out[12] = (int32_t)a.localSizeBytes;
	}
	if (n >= 15) {                 // search + narrow step in one launch (runs beside the bulk update like the two it replaces)
		hipFuncAttributes a{};
		HIPCHK(hipFuncGetAttributes(&a, (const void *)k_block_fast_narrow));
		out[13] = a.numRegs; out[14] = (int32_t)a.sharedSizeBytes;
	}
	if (n >= 17) {                 // round 5: the sparse block search (first pool size: beside the bulk update)
		hipFuncAttributes a{};
		HIPCHK(hipFuncGetAttributes(&a, (const void *)k_block_sparse<256, 4>));
		out[15] = a.numRegs; out[16] = (int32_t)a.sharedSizeBytes;
	}
	return GF2BV_OK;
	});
}

int gf2bv_stream_ceiling_device(int device, int64_t bytes, double *rmw_gbs, double *read_gbs)
{
	if (!rmw_gbs || !read_gbs || bytes < (1 << 20)) return fail(GF2BV_ERR_ARG, "bad ceiling request");
	int rc = check_device(device);
	if (rc) return rc;
	const int wgs = 2048;
	const i64 per = bytes / 16 / wgs;
	uint4 *buf = nullptr;
	unsigned *sink = nullptr;
	hipError_t e = hipMalloc(&buf, (size_t)per * wgs * 16);
	if (e != hipSuccess) return fail(GF2BV_ERR_NOMEM, "hipMalloc", e);
	HIPCHK(hipMalloc(&sink, 64));
	HIPCHK(hipMemset(buf, 1, (size_t)per * wgs * 16));
	hipEvent_t e0, e1;
	HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
	const int reps = 5;
	float ms = 0;
	k_rmw_stream<<<dim3(wgs), dim3(1024)>>>(buf, per, 5u);
	HIPCHK(hipEventRecord(e0, nullptr));
	for (int r = 0; r < reps; r++) k_rmw_stream<<<dim3(wgs), dim3(1024)>>>(buf, per, 5u);
	HIPCHK(hipEventRecord(e1, nullptr));
	HIPCHK(hipEventSynchronize(e1));
	HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	*rmw_gbs = 2.0 * (double)per * wgs * 16 * reps / (ms * 1e-3) / 1e9;
	k_read_stream<<<dim3(wgs), dim3(1024)>>>(buf, per, sink);
	HIPCHK(hipEventRecord(e0, nullptr));
	for (int r = 0; r < reps; r++) k_read_stream<<<dim3(wgs), dim3(1024)>>>(buf, per, sink);
	HIPCHK(hipEventRecord(e1, nullptr));
	HIPCHK(hipEventSynchronize(e1));
	HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	*read_gbs = (double)per * wgs * 16 * reps / (ms * 1e-3) / 1e9;
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	(void)hipFree(buf); (void)hipFree(sink);
	return GF2BV_OK;
}

int gf2bv_lds_clock_device(int device, double *shader_mhz, double *lds_bytes_per_clk_cu)
{
	if (!shader_mhz || !lds_bytes_per_clk_cu) return fail(GF2BV_ERR_ARG, "null pointer");
	int rc = check_device(device);
	if (rc) return rc;
	unsigned long long *d = nullptr, h[3] = { 0, 0, 0 };
	unsigned *sink = nullptr;
	struct Free { unsigned long long *&d; unsigned *&sink; ~Free() { pool().release(d); pool().release(sink); } } guard{ d, sink };     // (error paths too)
	HIPCHK(pool().alloc((void **)&d, sizeof h, device));
	HIPCHK(pool().alloc((void **)&sink, 64, device));
	int cus = 256;
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
	k_lds_clock<<<dim3(cus), dim3(512)>>>(d, 2000, sink);                   // warm-up (clocks ramp)
	k_lds_clock<<<dim3(cus), dim3(512)>>>(d, 40000, sink);                  // ~5 ms of ds_read_b128 on every CU
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
	if (!h[1]) return fail(GF2BV_ERR_HIP, "the clock probe measured nothing");
	*shader_mhz = (double)h[0] / ((double)h[1] / 100.0);                    // real-time counter: 100 MHz
	*lds_bytes_per_clk_cu = (double)h[2] * 1024.0 / (double)h[0];           // a ds_read_b128 wave-instruction moves 1 KiB
	return GF2BV_OK;
}

int gf2bv_device_alloc(int device, int64_t bytes, void **d_ptr)
{
	if (!d_ptr || bytes < 0) return fail(GF2BV_ERR_ARG, "bad alloc request");
	int rc = check_device(device);
	if (rc) return rc;
	// (not pooled: the buffer is the caller's; but a request the device cannot serve while the pool sits on idle buffers
	// gets them first)
	hipError_t e = pool().malloc_retry(d_ptr, (size_t)std::max<i64>(bytes, 16), device);
	if (e != hipSuccess) return fail(GF2BV_ERR_NOMEM, "hipMalloc", e);
	return GF2BV_OK;
}
// ---- pinned host staging for bindings (round 5) ------------------------------------------------------------------------------
// A binding that has to assemble its input on the host (the CPython shim copies every equation's ob_digit array into one buffer)
// gets that buffer here: page-locked, so the host-to-device copy of gf2bv_solve_digits is ONE DMA instead of the runtime's
// chunk-by-chunk staging of pageable memory, and recycled between calls (no page faults on a fresh 50 MB allocation every call).
// Up to four idle buffers are kept, GF2BV_HOST_POOL_MB MiB in all at most (default 4608: the two 2 GiB chunk buffers a batched
// list-of-int call alternates between -- page-locking 800 MB anew on every call was 200 of the 270 ms of a 16-system MT19937
// batch, profiles/r05_mt_many.txt); larger ones are freed on return, gf2bv_host_pool_trim frees the idle ones.
namespace {
struct HostPool {
	std::mutex mu;
	struct Buf { void *p; size_t bytes; };
	std::vector<Buf> idle;
	std::unordered_map<void *, size_t> live;
};
HostPool *host_pool_ptr() { static HostPool *p = new HostPool(); return p; }
}
int gf2bv_host_alloc(int64_t bytes, void **h_ptr)
{
	if (!h_ptr || bytes < 0) return fail(GF2BV_ERR_ARG, "bad alloc request");
	*h_ptr = nullptr;
	const size_t need = std::max<size_t>((size_t)bytes, 64);
	HostPool &P = *host_pool_ptr();
	{
		std::lock_guard<std::mutex> lk(P.mu);
		for (size_t i = 0; i < P.idle.size(); i++)
			if (P.idle[i].bytes >= need && P.idle[i].bytes <= 2 * need + ((size_t)1 << 20)) {
				*h_ptr = P.idle[i].p; P.live[*h_ptr] = P.idle[i].bytes;
				P.idle.erase(P.idle.begin() + i);
				return GF2BV_OK;
			}
	}
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(GF2BV_ERR_NODEVICE, "no HIP device visible");
	void *p = nullptr;
	const size_t cap = need + need / 8;
	hipError_t e = hipHostMalloc(&p, cap, hipHostMallocDefault);
	if (e != hipSuccess) { (void)hipGetLastError(); return fail(GF2BV_ERR_NOMEM, "hipHostMalloc", e); }
	std::lock_guard<std::mutex> lk(P.mu);
	P.live[p] = cap;
	*h_ptr = p;
	return GF2BV_OK;
}
void gf2bv_host_free(void *h_ptr)
{
	if (!h_ptr) return;
	HostPool &P = *host_pool_ptr();
	void *drop = nullptr;
	{
		std::lock_guard<std::mutex> lk(P.mu);
		auto it = P.live.find(h_ptr);
		if (it == P.live.end()) return;
		const size_t bytes = it->second;
		P.live.erase(it);
		size_t kept = 0;
		for (const auto &b : P.idle) kept += b.bytes;
		size_t cap_mb = 4608;
		if (const char *e = getenv("GF2BV_HOST_POOL_MB")) { long v = atol(e); if (v >= 0) cap_mb = (size_t)v; }
		if (P.idle.size() < 4 && kept + bytes <= (cap_mb << 20)) P.idle.push_back({ h_ptr, bytes });
		else drop = h_ptr;
	}
	if (drop) (void)hipHostFree(drop);
}

int64_t gf2bv_host_pool_trim(void)
{
	HostPool &P = *host_pool_ptr();
	std::vector<HostPool::Buf> drop;
	{
		std::lock_guard<std::mutex> lk(P.mu);
		drop.swap(P.idle);
	}
	int64_t bytes = 0;
	for (const auto &b : drop) { bytes += (int64_t)b.bytes; (void)hipHostFree(b.p); }
	return bytes;
}

// The gang size gf2bv_solve_batch_* would choose for nsys systems of rows x cols with free_bytes of device memory free -- a pure
// function (no device is touched): bench.py --dry-run-ranks prints every rank's plan of the multi-GPU batch job with it.
int64_t gf2bv_plan_gang(int64_t nsys, int64_t rows, int64_t cols, int64_t free_bytes)
{
	if (nsys <= 0 || rows <= 0 || cols <= 0 || free_bytes < 0) return 0;
	return pick_gang(nsys, rows, cols, free_bytes);
}

int64_t gf2bv_pool_trim(int device)
{
	if (check_device(device)) return -1;
	return (int64_t)pool().trim(device);
}
int64_t gf2bv_pool_idle_bytes(int device)
{
	if (device < 0) return -1;
	Pool &P = pool();
	std::lock_guard<std::mutex> lk(P.mu);
	size_t bytes = 0;
	auto bf = P.big_free.find(device);
	if (bf != P.big_free.end()) for (const Pool::Big &b : bf->second) bytes += b.bytes;
	for (auto &kv : P.free_bufs) if (kv.first.first == device) bytes += kv.first.second * kv.second.size();
	return (int64_t)bytes;
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

#ifdef GF2_SPARSE_DEBUG
extern "C" int gf2bv_sparse_probe_read(unsigned long long *w, int reset)
{
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpyFromSymbol(w, HIP_SYMBOL(gf2_sparse_probe), sizeof(gf2_sparse_probe)));
	if (reset) { unsigned long long z[8] = { 0 }; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(gf2_sparse_probe), z, sizeof z)); }
	return GF2BV_OK;
}
#endif
#ifdef GF2_STEP_PROBE
// Probe build only (tools/probe_step.py): choose the block whose panel steps are recorded, fetch the records.
int gf2bv_probe_set(int j0)
{
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(gf2_probe_j0), &j0, sizeof(int)));
	void *p = nullptr;
	HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(gf2_probe_upd))); HIPCHK(hipMemset(p, 0, sizeof(gf2_probe_upd)));
	HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(gf2_probe_wg))); HIPCHK(hipMemset(p, 0, sizeof(gf2_probe_wg)));
	HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(gf2_probe_un))); HIPCHK(hipMemset(p, 0, sizeof(gf2_probe_un)));
	return GF2BV_OK;
}
int gf2bv_probe_read_update(unsigned long long *upd)
{
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpyFromSymbol(upd, HIP_SYMBOL(gf2_probe_upd), sizeof(gf2_probe_upd)));
	return GF2BV_OK;
}
int gf2bv_probe_read_gj(unsigned long long *w)
{
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpyFromSymbol(w, HIP_SYMBOL(gf2_probe_gj), sizeof(gf2_probe_gj)));
	return GF2BV_OK;
}
int gf2bv_probe_read_fast(unsigned long long *w)
{
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpyFromSymbol(w, HIP_SYMBOL(gf2_probe_fast), sizeof(gf2_probe_fast)));
	return GF2BV_OK;
}
int gf2bv_probe_read_wave(unsigned long long *w)
{
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpyFromSymbol(w, HIP_SYMBOL(gf2_probe_wave), sizeof(gf2_probe_wave)));
	return GF2BV_OK;
}
int gf2bv_probe_read(unsigned long long *wg, unsigned long long *un)
{
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpyFromSymbol(wg, HIP_SYMBOL(gf2_probe_wg), sizeof(gf2_probe_wg)));
	HIPCHK(hipMemcpyFromSymbol(un, HIP_SYMBOL(gf2_probe_un), sizeof(gf2_probe_un)));
	return GF2BV_OK;
}
#endif

}  // extern "C"
