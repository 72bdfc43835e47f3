// _internal.cpp -- CPython extension `gf2bv_amd._internal`.
//
// Mirrors the native boundary of the reference (maple3142/gf2bv, gf2bv/_internal.c:767-833):
// same callables (m4ri_solve, to_bits, xor_tuple, tuple_where, mul_bit_quad), same types
// (AffineSpace, AffineSpaceIterator, AffineSpaceIteratorSlow), same argument meaning and
// error behaviour -- but m4ri_solve hands the equations to libgf2bv_hip.so (HIP kernels on
// MI355X) instead of M4RI.  No GF(2) elimination happens in this file and there is no CPU
// fallback: without a GPU m4ri_solve raises RuntimeError.
//
// Host-side work kept here, as in the reference: reading CPython int digits
// (_internal.c:5-16, 41-59 -- here a memcpy of ob_digit, the bit shuffling is the device pack
// kernel), converting word vectors back to ints (_internal.c:32-39) and the tiny
// origin ^ basis[i] combinations of AffineSpace.get / iteration (_internal.c:101-122, 242-273).
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/gf2bv_hip.h"

#if PY_VERSION_HEX >= 0x030C0000
#define GF2_DIGIT_COUNT(o) ((Py_ssize_t)((o)->long_value.lv_tag >> 3))
#define GF2_DIGITS(o) ((o)->long_value.ob_digit)
#else
#define GF2_DIGIT_COUNT(o) (Py_SIZE(o) < 0 ? -Py_SIZE(o) : Py_SIZE(o))
#define GF2_DIGITS(o) ((o)->ob_digit)
#endif

namespace {

PyObject *words_to_pylong(const uint64_t *w, int64_t nwords)
{
	if (nwords <= 0) return PyLong_FromLong(0);
	return _PyLong_FromByteArray(reinterpret_cast<const unsigned char *>(w), (size_t)nwords * 8, 1, 0);
}

// bit `i` (LSB = 0) of |v|, false beyond its digits
inline bool long_bit(PyLongObject *v, Py_ssize_t i)
{
	Py_ssize_t d = i / PyLong_SHIFT;
	if (d >= GF2_DIGIT_COUNT(v)) return false;
	return (GF2_DIGITS(v)[d] >> (i % PyLong_SHIFT)) & 1;
}

// ------------------------------------------------------------------------------------------------
// AffineSpace: origin + basis as 64-bit word vectors (the reference keeps two mzd_t*,
// gf2bv/_internal.h:7-10).
struct AffineSpaceObject {
	PyObject_HEAD
	int64_t dim, words;
	uint64_t *origin;   // words
	uint64_t *basis;    // dim x words
	int device;         // the device whose solve produced the space (its large walks are materialised there);
	                    // -1: built from host integers by the test hook _space_from_ints -- walked on the host
};

struct SpaceIterObject {
	PyObject_HEAD
	AffineSpaceObject *space;
	uint64_t *cur;      // Gray: running vector; Slow: scratch
	uint64_t idx;       // Gray index
	uint8_t *state;     // Slow: dim+1 counter bits
	int done;
	// Large spaces: the elements are materialised on the device a chunk at a time (gf2bv_space_enumerate) and the
	// iterator only slices ints out of the chunk -- no per-item XOR on the host.  Small ones (a chunk would be under
	// 64 KiB) are stepped on the host as in the reference: a kernel launch costs more than their whole walk.
	gf2bv_space *dev;   // nullptr: host stepping
	const uint64_t *chunk;   // the space handle's pinned buffer, filled by the last gf2bv_space_enumerate
	int64_t chunk_cap, chunk_fill, chunk_pos;
	uint64_t next_first;     // index of the first element of the next chunk
	int wrapped;             // next_first has run past 2^64 - 1 (dimension 64 only)
};

constexpr int64_t kChunkBytes = 8 << 20, kDeviceMinBytes = 64 << 10;

PyTypeObject *AffineSpace_Type, *SpaceIterGray_Type, *SpaceIterSlow_Type;

void space_dealloc(AffineSpaceObject *self)
{
	PyTypeObject *tp = Py_TYPE(self);
	free(self->origin);
	free(self->basis);
	tp->tp_free((PyObject *)self);
	Py_DECREF(tp);
}

PyObject *space_dimension(AffineSpaceObject *self, void *) { return PyLong_FromLongLong(self->dim); }
PyObject *space_device(AffineSpaceObject *self, void *) { return PyLong_FromLong(self->device); }
PyObject *space_origin(AffineSpaceObject *self, void *) { return words_to_pylong(self->origin, self->words); }
PyObject *space_basis(AffineSpaceObject *self, void *)
{
	PyObject *t = PyTuple_New(self->dim);   // a tuple, like _internal.c:218-230 (the .pyi says list)
	if (!t) return nullptr;
	for (int64_t i = 0; i < self->dim; i++) {
		PyObject *v = words_to_pylong(self->basis + i * self->words, self->words);
		if (!v) { Py_DECREF(t); return nullptr; }
		PyTuple_SET_ITEM(t, i, v);
	}
	return t;
}

// AffineSpace.get(n): origin ^ XOR of basis rows at the set bits of n (binary, not Gray; only
// the low `dimension` bits of |n| are read) -- _internal.c:242-273
PyObject *space_get(AffineSpaceObject *self, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 1) { PyErr_SetString(PyExc_TypeError, "get requires 1 argument"); return nullptr; }
	if (!PyLong_Check(args[0])) { PyErr_SetString(PyExc_TypeError, "Index must be an integer"); return nullptr; }
	PyLongObject *n = (PyLongObject *)args[0];
	int64_t sw = (self->dim + 63) / 64;
	std::vector<uint64_t> sel((size_t)(sw ? sw : 1), 0), out((size_t)(self->words ? self->words : 1), 0);
	for (int64_t i = 0; i < self->dim; i++)
		if (long_bit(n, i)) sel[i >> 6] |= (uint64_t)1 << (i & 63);
	gf2bv_space_combine(self->origin, self->basis, self->dim, self->words, sel.data(), sw, out.data());
	return words_to_pylong(out.data(), self->words);
}

PyObject *space_iter(AffineSpaceObject *self)
{
	const bool gray = self->dim <= 64;       // _internal.c:185
	PyTypeObject *tp = gray ? SpaceIterGray_Type : SpaceIterSlow_Type;
	SpaceIterObject *it = (SpaceIterObject *)tp->tp_alloc(tp, 0);
	if (!it) return nullptr;
	Py_INCREF(self);
	it->space = self;
	it->idx = 0;
	it->done = 0;
	it->state = nullptr;
	size_t nw = (size_t)(self->words ? self->words : 1);
	it->dev = nullptr; it->chunk = nullptr; it->chunk_cap = it->chunk_fill = it->chunk_pos = 0; it->next_first = 0; it->wrapped = 0;
	it->cur = (uint64_t *)malloc(nw * sizeof(uint64_t));
	if (!gray) it->state = (uint8_t *)calloc((size_t)self->dim + 1, 1);
	if (!it->cur || (!gray && !it->state)) { Py_DECREF(it); return PyErr_NoMemory(); }
	memcpy(it->cur, self->origin, (size_t)self->words * sizeof(uint64_t));
	// device enumeration when a chunk is worth a launch: min(2^dim, chunk capacity) elements of `words` words
	const int64_t row_bytes = (int64_t)nw * 8;
	int64_t cap = std::max<int64_t>(1, kChunkBytes / row_bytes);
	if (self->dim < 40) cap = std::min<int64_t>(cap, (int64_t)1 << self->dim);
	if (self->device >= 0 && self->words > 0 && cap * row_bytes >= kDeviceMinBytes) {
		if (gf2bv_space_open(self->origin, self->basis, self->dim, self->words, self->device, &it->dev) != GF2BV_OK) {
			PyErr_SetString(PyExc_RuntimeError, gf2bv_last_error());
			Py_DECREF(it);
			return nullptr;
		}
		it->chunk_cap = cap;
	}
	return (PyObject *)it;
}

// total number of elements still to be produced from index `first` on, capped at `cap` (2^dim may not fit 64 bits)
int64_t remaining_from(int64_t dim, uint64_t first, int wrapped, int64_t cap)
{
	if (wrapped) return 0;
	if (dim >= 64) {                                   // 2^64 (or more) elements: only the wrap ends the walk
		const uint64_t left = ~first;                  // elements first .. 2^64 - 1 = left + 1
		return (left >= (uint64_t)cap) ? cap : (int64_t)left + 1;
	}
	const uint64_t total = (uint64_t)1 << dim;
	if (first >= total) return 0;
	return (int64_t)std::min<uint64_t>(total - first, (uint64_t)cap);
}

// next element from the device-filled chunk; refills when it runs dry.  Returns nullptr with no error set at the end.
PyObject *iter_next_chunked(SpaceIterObject *self, int gray)
{
	AffineSpaceObject *sp = self->space;
	if (self->chunk_pos >= self->chunk_fill) {
		// (binary walk with dimension > 64: the first 2^64 elements only involve basis[0..63]; nobody gets further)
		const int64_t n = remaining_from(sp->dim, self->next_first, self->wrapped, self->chunk_cap);
		if (n <= 0) return nullptr;
		int rc;
		Py_BEGIN_ALLOW_THREADS
		rc = gf2bv_space_enumerate(self->dev, self->next_first, n, gray, nullptr);
		Py_END_ALLOW_THREADS
		if (rc != GF2BV_OK) { PyErr_SetString(PyExc_RuntimeError, gf2bv_last_error()); return nullptr; }
		self->chunk = gf2bv_space_buffer(self->dev);
		const uint64_t before = self->next_first;
		self->next_first += (uint64_t)n;
		if (self->next_first < before || (sp->dim >= 64 && self->next_first == 0)) self->wrapped = 1;
		self->chunk_fill = n; self->chunk_pos = 0;
	}
	return words_to_pylong(self->chunk + self->chunk_pos++ * sp->words, sp->words);
}

void iter_dealloc(SpaceIterObject *self)
{
	PyTypeObject *tp = Py_TYPE(self);
	Py_XDECREF(self->space);
	if (self->dev) gf2bv_space_close(self->dev);
	free(self->cur);
	free(self->state);
	tp->tp_free((PyObject *)self);
	Py_DECREF(tp);
}

// Reflected Gray code walk, _internal.c:101-122: yield the running vector, then flip the basis
// row whose index is the bit in which gray(idx) and gray(idx+1) differ.
PyObject *iter_next_gray(SpaceIterObject *self)
{
	if (self->dev) return iter_next_chunked(self, 1);
	if (self->done) return nullptr;
	AffineSpaceObject *sp = self->space;
	PyObject *ret = words_to_pylong(self->cur, sp->words);
	uint64_t x = self->idx ^ (self->idx >> 1);
	self->idx++;
	uint64_t y = self->idx ^ (self->idx >> 1);
	int diff = __builtin_ctzll(x ^ y);
	if (diff >= sp->dim || (sp->dim == 64 && self->idx == 0)) {
		self->done = 1;
		return ret;
	}
	const uint64_t *b = sp->basis + (int64_t)diff * sp->words;
	for (int64_t w = 0; w < sp->words; w++) self->cur[w] ^= b[w];
	return ret;
}

// Binary counter walk for dimension > 64, _internal.c:63-91 (basis[0] is the LSB).
PyObject *iter_next_slow(SpaceIterObject *self)
{
	if (self->dev) return iter_next_chunked(self, 0);
	AffineSpaceObject *sp = self->space;
	const int64_t n = sp->dim;
	if (self->state[n]) return nullptr;
	memcpy(self->cur, sp->origin, (size_t)sp->words * sizeof(uint64_t));
	for (int64_t r = 0; r < n; r++)
		if (self->state[r])
			for (int64_t w = 0; w < sp->words; w++) self->cur[w] ^= sp->basis[r * sp->words + w];
	uint8_t sentinel = 1;
	for (int64_t r = 0; r < n; r++) {
		self->state[r] ^= 1;
		if (self->state[r]) { sentinel = 0; break; }
	}
	self->state[n] = sentinel;
	return words_to_pylong(self->cur, sp->words);
}

PyGetSetDef space_getset[] = {
	{"dimension", (getter)space_dimension, nullptr, "Dimension of the affine space", nullptr},
	{"origin", (getter)space_origin, nullptr, "Origin of the affine space", nullptr},
	{"basis", (getter)space_basis, nullptr, "Basis of the affine space", nullptr},
	{"device", (getter)space_device, nullptr, "GPU whose solve produced the space (-1: built on the host); not in the reference", nullptr},
	{nullptr, nullptr, nullptr, nullptr, nullptr}};

PyMethodDef space_methods[] = {
	{"get", (PyCFunction)(void (*)(void))space_get, METH_FASTCALL,
	 "get(n)\n--\n\nGet the n-th element of the affine space, should check 0 <= n < 2**(space.dimension) first."},
	{nullptr, nullptr, 0, nullptr}};

PyType_Slot space_slots[] = {
	{Py_tp_dealloc, (void *)space_dealloc}, {Py_tp_iter, (void *)space_iter},
	{Py_tp_methods, (void *)space_methods}, {Py_tp_getset, (void *)space_getset}, {0, nullptr}};
PyType_Slot gray_slots[] = {
	{Py_tp_dealloc, (void *)iter_dealloc}, {Py_tp_iter, (void *)PyObject_SelfIter},
	{Py_tp_iternext, (void *)iter_next_gray}, {0, nullptr}};
PyType_Slot slow_slots[] = {
	{Py_tp_dealloc, (void *)iter_dealloc}, {Py_tp_iter, (void *)PyObject_SelfIter},
	{Py_tp_iternext, (void *)iter_next_slow}, {0, nullptr}};

#ifndef Py_TPFLAGS_DISALLOW_INSTANTIATION
#define Py_TPFLAGS_DISALLOW_INSTANTIATION 0
#endif
PyType_Spec space_spec = {"_internal.AffineSpace", sizeof(AffineSpaceObject), 0,
                          Py_TPFLAGS_DEFAULT | Py_TPFLAGS_DISALLOW_INSTANTIATION, space_slots};
PyType_Spec gray_spec = {"_internal.AffineSpaceIterator", sizeof(SpaceIterObject), 0,
                         Py_TPFLAGS_DEFAULT | Py_TPFLAGS_DISALLOW_INSTANTIATION, gray_slots};
PyType_Spec slow_spec = {"_internal.AffineSpaceIteratorSlow", sizeof(SpaceIterObject), 0,
                         Py_TPFLAGS_DEFAULT | Py_TPFLAGS_DISALLOW_INSTANTIATION, slow_slots};

// Result handle -> None / int / AffineSpace (_internal.c:440-501); frees the handle.
PyObject *result_to_py(gf2bv_result *res, long mode, int device)
{
	if (gf2bv_result_status(res) != GF2BV_STATUS_SOLVED) {
		gf2bv_result_free(res);
		Py_RETURN_NONE;              // inconsistent system -> None (_internal.c:440-446)
	}
	const int64_t words = gf2bv_result_words(res);
	PyObject *ret = nullptr;
	if (mode == GF2BV_MODE_SINGLE) {
		std::vector<uint64_t> o((size_t)(words ? words : 1));
		gf2bv_result_origin(res, o.data());
		ret = words_to_pylong(o.data(), words);
	} else {
		AffineSpaceObject *sp = (AffineSpaceObject *)AffineSpace_Type->tp_alloc(AffineSpace_Type, 0);
		if (sp) {
			sp->dim = gf2bv_result_dimension(res);
			sp->words = words;
			sp->device = device;
			sp->origin = (uint64_t *)calloc((size_t)(words ? words : 1), sizeof(uint64_t));
			sp->basis = (uint64_t *)calloc((size_t)((sp->dim * words) > 0 ? sp->dim * words : 1), sizeof(uint64_t));
			if (!sp->origin || !sp->basis) {
				Py_DECREF(sp);                   // (space_dealloc frees whichever half exists)
				gf2bv_result_free(res);
				return PyErr_NoMemory();
			}
			gf2bv_result_origin(res, sp->origin);
			gf2bv_result_basis(res, sp->basis);
			ret = (PyObject *)sp;
		}
	}
	gf2bv_result_free(res);
	return ret;
}

// Which GPU a solve runs on.  The reference's m4ri_solve has no such notion (one core of the host); here every solve entry
// takes an OPTIONAL trailing `device` argument (an index below device_count()), and without it uses the module default
// (set_default_device; initially the environment variable GF2BV_DEVICE, else 0).  AffineSpace remembers the device of
// its solve: large walks are materialised there.
int g_default_device = -1;
int default_device()
{
	if (g_default_device < 0) {
		const char *e = getenv("GF2BV_DEVICE");
		g_default_device = e ? atoi(e) : 0;
		if (g_default_device < 0) g_default_device = 0;
	}
	return g_default_device;
}
bool parse_device(PyObject *obj, int *device)
{
	if (obj == Py_None) { *device = default_device(); return true; }
	const long d = PyLong_AsLong(obj);
	if (d == -1 && PyErr_Occurred()) return false;
	const int n = gf2bv_device_count();
	if (d < 0 || (n > 0 && d >= n)) { PyErr_Format(PyExc_ValueError, "device %ld out of range (%d visible)", d, n); return false; }
	*device = (int)d;
	return true;
}
// devices: None = the module's default device (set_default_device / GF2BV_DEVICE -- what m4ri_solve and m4ri_solve_packed use:
// a process pinned to one GPU stays on it), the string "all" = every visible device, an int, or a sequence of ints (repeats allowed)
bool parse_devices(PyObject *obj, std::vector<int> *devs)
{
	devs->clear();
	if (obj == Py_None) { devs->push_back(default_device()); return true; }
	if (PyUnicode_Check(obj)) {
		if (PyUnicode_CompareWithASCIIString(obj, "all") != 0) {
			PyErr_SetString(PyExc_ValueError, "devices must be None, 'all', an int or a sequence of ints");
			return false;
		}
		const int n = gf2bv_device_count();
		for (int d = 0; d < (n > 0 ? n : 1); d++) devs->push_back(d);
		return true;
	}
	if (PyLong_Check(obj)) { int d; if (!parse_device(obj, &d)) return false; devs->push_back(d); return true; }
	PyObject *seq = PySequence_Fast(obj, "devices must be None, 'all', an int or a sequence of ints");
	if (!seq) return false;
	const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
	for (Py_ssize_t i = 0; i < n; i++) {
		int d;
		if (!parse_device(PySequence_Fast_GET_ITEM(seq, i), &d)) { Py_DECREF(seq); return false; }
		devs->push_back(d);
	}
	Py_DECREF(seq);
	if (devs->empty()) { PyErr_SetString(PyExc_ValueError, "devices must not be empty"); return false; }
	return true;
}

// cols / mode checks shared by m4ri_solve and m4ri_solve_many (_internal.c:372-395)
bool parse_cols_mode(PyObject *cols_obj, PyObject *mode_obj, Py_ssize_t *cols, long *mode)
{
	*cols = PyLong_AsSsize_t(cols_obj);
	if (*cols <= 0) {
		if (*cols == -1 && PyErr_Occurred()) return false;
		PyErr_SetString(PyExc_ValueError, "Number of columns must be positive");
		return false;
	}
	*mode = PyLong_AsLong(mode_obj);
	if (*mode == -1 && PyErr_Occurred()) return false;
	if (*mode != GF2BV_MODE_SINGLE && *mode != GF2BV_MODE_AFFINE_SPACE) {
		PyErr_SetString(PyExc_ValueError, "Invalid mode");
		return false;
	}
	return true;
}

// The digits that can hold bits 0..cols of every equation (sign ignored, higher bits ignored), gathered into ONE
// buffer for the C ABI.  Two phases: offsets + source pointers row by row (type check), then the copy -- into memory
// that is NOT value-initialised first, and by several threads when it is large (the GIL is held by the caller,
// nothing can change the ints; the workers only read): for the 20000 x 19968 recovery systems of the reference's
// examples (53 MB of digits) a zero-filled std::vector plus a single-threaded memcpy were 10 of the 24 ms of a solve.
struct Staging {
	uint32_t *digits = nullptr;
	bool pinned = false;                                       // digits came from gf2bv_host_alloc (page-locked, recycled)
	void drop() { if (pinned) gf2bv_host_free(digits); else free(digits); digits = nullptr; pinned = false; }
	~Staging() { drop(); }
};
struct DigitGather {
	std::vector<int64_t> off = std::vector<int64_t>(1, 0);    // off[r] .. off[r+1]: digits of row r
	std::vector<const uint32_t *> src;                         // first digit of row r
	Staging own;
	uint32_t *digits = nullptr;                                // (gather(): every row, in `own`)
	// `src` points into the ints' ob_digit arrays.  m4ri_solve gathers before it first drops the GIL, so borrowed pointers do;
	// m4ri_solve_many gathers chunk k + 1 AFTER it dropped the GIL to wait for chunk k - 1 -- another Python thread may have
	// emptied the caller's lists by then -- and therefore holds a strong reference to every int until the call ends.
	bool hold = false;
	std::vector<PyObject *> held;
	DigitGather() = default;
	DigitGather(const DigitGather &) = delete;
	DigitGather &operator=(const DigitGather &) = delete;
	~DigitGather() { for (PyObject *o : held) Py_DECREF(o); }     // (destroyed by the Python-facing function, GIL held)
	bool add(PyObject *list, Py_ssize_t cols)
	{
		const Py_ssize_t need = (cols + 1 + PyLong_SHIFT - 1) / PyLong_SHIFT;
		const Py_ssize_t rows = PyList_GET_SIZE(list);
		static_assert(sizeof(digit) == sizeof(uint32_t), "30-bit digits in uint32 expected");
		for (Py_ssize_t r = 0; r < rows; r++) {
			PyObject *item = PyList_GET_ITEM(list, r);
			if (!PyLong_Check(item)) {
				PyErr_SetString(PyExc_TypeError, "List items must be integers");
				return false;
			}
			const Py_ssize_t nd = GF2_DIGIT_COUNT((PyLongObject *)item);
			off.push_back(off.back() + (nd < need ? nd : need));
			src.push_back(reinterpret_cast<const uint32_t *>(GF2_DIGITS((PyLongObject *)item)));
			if (hold) { Py_INCREF(item); held.push_back(item); }
		}
		return true;
	}
	bool gather()
	{
		if (!gather_rows(0, src.size(), (size_t)1 << 30, own)) return false;
		digits = own.digits;
		return true;
	}
	// Rows r0 .. r1 into a buffer of this object's own (row r's digits at digits + off[r] - off[r0]).  Page-locked staging
	// from the library when the copy is worth a DMA of its own and not larger than pin_max (page-locking more than a call can
	// recycle costs more than the staged copy it saves); no GPU, or no pinned memory left: malloc.
	bool gather_rows(size_t r0, size_t r1, size_t pin_max, Staging &st) const
	{
		st.drop();
		uint32_t *digits = nullptr;
		const int64_t base = off[r0];
		const size_t total = (size_t)(off[r1] - base);
		void *hp = nullptr;
		if ((total + 1) * sizeof(uint32_t) >= (256u << 10) && (total + 1) * sizeof(uint32_t) <= pin_max && gf2bv_host_alloc((int64_t)((total + 1) * sizeof(uint32_t)), &hp) == GF2BV_OK && hp) {
			digits = static_cast<uint32_t *>(hp); st.pinned = true;
		} else digits = static_cast<uint32_t *>(malloc((total + 1) * sizeof(uint32_t)));
		if (!digits) { PyErr_NoMemory(); return false; }
		st.digits = digits;
		auto copy_rows = [this, base, digits](size_t a, size_t b) {
			for (size_t r = a; r < b; r++)
				memcpy(digits + (off[r] - base), src[r], (size_t)(off[r + 1] - off[r]) * sizeof(uint32_t));
		};
		unsigned nt = total * sizeof(uint32_t) >= (8u << 20) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
		if (nt <= 1 || r1 - r0 < 64) { copy_rows(r0, r1); return true; }
		// equal shares of the DIGITS, not of the rows
		std::vector<std::thread> th;
		size_t a = r0;
		for (unsigned k = 1; k <= nt; k++) {
			size_t b = r1;
			if (k < nt) {
				const int64_t want = base + (int64_t)(total / nt * k);
				b = (size_t)(std::lower_bound(off.begin() + (ptrdiff_t)r0, off.begin() + (ptrdiff_t)r1 + 1, want) - off.begin());
				if (b > r1) b = r1;
				if (b < a) b = a;
			}
			if (k < nt) th.emplace_back(copy_rows, a, b); else copy_rows(a, b);
			a = b;
		}
		for (auto &t : th) t.join();
		return true;
	}
};

// ------------------------------------------------------------------------------------------------
// m4ri_solve(equations, cols, mode) -- gf2bv/_internal.c:359-502
PyObject *py_m4ri_solve(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 3 && nargs != 4) { PyErr_SetString(PyExc_TypeError, "m4ri_solve requires 3 arguments"); return nullptr; }
	int device = default_device();
	if (nargs == 4 && !parse_device(args[3], &device)) return nullptr;
	PyObject *list = args[0];
	if (!PyList_Check(list)) {
		PyErr_SetString(PyExc_TypeError, "The first argument equations must be a list");
		return nullptr;
	}
	Py_ssize_t cols;
	long mode;
	if (!parse_cols_mode(args[1], args[2], &cols, &mode)) return nullptr;
	const Py_ssize_t rows = PyList_GET_SIZE(list);
	if (rows < cols) {
		PyErr_SetString(PyExc_ValueError,
		                "Number of rows must be greater than or equal to number of columns, try pad with zeros.");
		return nullptr;
	}
	DigitGather dg;
	dg.off.reserve((size_t)rows + 1); dg.src.reserve((size_t)rows);
	if (!dg.add(list, cols) || !dg.gather()) return nullptr;

	gf2bv_result *res = nullptr;
	int rc;
	Py_BEGIN_ALLOW_THREADS          // same place the reference drops the GIL (_internal.c:429)
	rc = gf2bv_solve_digits(dg.digits, dg.off.data(), PyLong_SHIFT, rows, cols, (int)mode, device, &res);
	Py_END_ALLOW_THREADS
	if (rc != GF2BV_OK) {
		PyErr_Format(rc == GF2BV_ERR_ARG ? PyExc_ValueError : PyExc_RuntimeError,
		             "gf2bv_amd: HIP solve failed (%d): %s", rc, gf2bv_last_error());
		return nullptr;
	}
	return result_to_py(res, mode, device);
}

// m4ri_solve_packed(buffer, rows, words, cols, mode) -> None | int | AffineSpace.
// New entry: the equations arrive ALREADY PACKED -- `rows` x `words` little-endian 64-bit words in the
// bit order of the equation ints (bit 0 = affine term, bit k = coefficient of variable k-1), e.g. the numpy array a
// PackedLinearSystem builds -- instead of as a list of Python ints.  No PyLong is created or read: the buffer goes
// to gf2bv_solve_digits as 32-bit "digits" and the device pack kernel (k_pack_digits) does what
// gf2bv/_internal.c:403-426 does bit by bit.  Same checks, same result types as m4ri_solve.
PyObject *py_m4ri_solve_packed(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 5 && nargs != 6) { PyErr_SetString(PyExc_TypeError, "m4ri_solve_packed requires 5 arguments"); return nullptr; }
	int device = default_device();
	if (nargs == 6 && !parse_device(args[5], &device)) return nullptr;
	const Py_ssize_t rows = PyLong_AsSsize_t(args[1]), words = PyLong_AsSsize_t(args[2]);
	if ((rows == -1 || words == -1) && PyErr_Occurred()) return nullptr;
	Py_ssize_t cols;
	long mode;
	if (!parse_cols_mode(args[3], args[4], &cols, &mode)) return nullptr;
	if (rows < cols) {
		PyErr_SetString(PyExc_ValueError,
		                "Number of rows must be greater than or equal to number of columns, try pad with zeros.");
		return nullptr;
	}
	Py_buffer view;
	if (PyObject_GetBuffer(args[0], &view, PyBUF_C_CONTIGUOUS) != 0) return nullptr;
	if (words <= 0 || view.len != rows * words * 8 || words * 64 < cols + 1) {
		PyBuffer_Release(&view);
		PyErr_SetString(PyExc_ValueError, "buffer must hold rows x words 64-bit words covering cols + 1 bits");
		return nullptr;
	}
	std::vector<int64_t> off;
	try { off.resize((size_t)rows + 1); } catch (const std::bad_alloc &) { PyBuffer_Release(&view); return PyErr_NoMemory(); }
	for (Py_ssize_t r = 0; r <= rows; r++) off[(size_t)r] = (int64_t)r * words * 2;
	gf2bv_result *res = nullptr;
	int rc;
	Py_BEGIN_ALLOW_THREADS
	rc = gf2bv_solve_digits(static_cast<const uint32_t *>(view.buf), off.data(), 32, rows, cols, (int)mode, device, &res);
	Py_END_ALLOW_THREADS
	PyBuffer_Release(&view);
	if (rc != GF2BV_OK) {
		PyErr_Format(rc == GF2BV_ERR_ARG ? PyExc_ValueError : PyExc_RuntimeError,
		             "gf2bv_amd: HIP solve failed (%d): %s", rc, gf2bv_last_error());
		return nullptr;
	}
	return result_to_py(res, mode, device);
}

// m4ri_solve_many(list_of_equation_lists, cols, mode[, devices]) -> list of (None | int | AffineSpace).
// New entry (no counterpart in the reference): independent systems of one shape -- one per output
// bit / per instance in the recovery examples -- are solved as lock-step gangs by one call; every
// element of the result is what m4ri_solve would return for that system.  `devices`: None (default) = the module's default
// device, "all" = every visible GPU, an int, or a sequence of device indices -- the systems are sharded in contiguous blocks over them,
// one host thread per entry inside the library (gf2bv_solve_batch_digits_multi), results in input order.
PyObject *py_m4ri_solve_many(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 3 && nargs != 4) { PyErr_SetString(PyExc_TypeError, "m4ri_solve_many requires 3 arguments"); return nullptr; }
	std::vector<int> devs;
	if (!parse_devices(nargs == 4 ? args[3] : Py_None, &devs)) return nullptr;
	PyObject *systems = args[0];
	if (!PyList_Check(systems)) {
		PyErr_SetString(PyExc_TypeError, "The first argument must be a list of equation lists");
		return nullptr;
	}
	Py_ssize_t cols;
	long mode;
	if (!parse_cols_mode(args[1], args[2], &cols, &mode)) return nullptr;
	const Py_ssize_t nsys = PyList_GET_SIZE(systems);
	if (nsys == 0) return PyList_New(0);
	Py_ssize_t rows = -1;
	DigitGather dg;
	dg.hold = true;                        // later chunks are gathered after the GIL was dropped: own the ints for the whole call
	for (Py_ssize_t s = 0; s < nsys; s++) {
		PyObject *list = PyList_GET_ITEM(systems, s);
		if (!PyList_Check(list)) {
			PyErr_SetString(PyExc_TypeError, "The first argument must be a list of equation lists");
			return nullptr;
		}
		if (rows < 0) rows = PyList_GET_SIZE(list);
		if (PyList_GET_SIZE(list) != rows) {
			PyErr_SetString(PyExc_ValueError, "All systems of a batch need the same number of rows, pad with zeros.");
			return nullptr;
		}
		if (rows < cols) {
			PyErr_SetString(PyExc_ValueError,
			                "Number of rows must be greater than or equal to number of columns, try pad with zeros.");
			return nullptr;
		}
		if (!dg.add(list, cols)) return nullptr;
	}
	// The digits go to the library in CHUNKS of whole systems, at most GF2BV_BATCH_CHUNK_MB MiB each (default 2048; an even
	// number of systems per chunk, so that the library still forms two gangs of it): chunk k + 1 is gathered (GIL held, as for
	// one system) while chunk k is being solved by a thread of its own, and the two page-locked buffers the chunks alternate
	// between are recycled by the library's staging pool.  Gathering ALL the digits first -- 16 MT19937 recovery systems are
	// 800 MB, 64 systems of 32768^2 are 8.6 GB -- meant page-locking (or page-faulting) that much on every call: 226 of the
	// 270 ms of the 16-system call, where the solves take 44 (profiles/r05_mt_many.txt).
	const size_t total_digits = (size_t)dg.off.back();
	size_t chunk_bytes = (size_t)2048 << 20;
	if (const char *e = getenv("GF2BV_BATCH_CHUNK_MB")) { long v = atol(e); if (v >= 1) chunk_bytes = (size_t)v << 20; }
	Py_ssize_t chunk_sys = nsys;
	if (total_digits * sizeof(uint32_t) > chunk_bytes) {
		const size_t per_sys = std::max<size_t>(1, total_digits * sizeof(uint32_t) / (size_t)nsys);
		chunk_sys = (Py_ssize_t)std::max<size_t>(1, chunk_bytes / per_sys);
		if (chunk_sys >= 2) chunk_sys &= ~(Py_ssize_t)1;
		// equal chunks
		const Py_ssize_t nchunks = (nsys + chunk_sys - 1) / chunk_sys;
		chunk_sys = (nsys + nchunks - 1) / nchunks;
		if (chunk_sys >= 2 && (chunk_sys & 1) && chunk_sys < nsys) chunk_sys++;
	}
	std::vector<gf2bv_result *> res((size_t)nsys, nullptr);
	std::vector<int> sysdev((size_t)nsys, devs[0]);
	struct Chunk {
		Staging st;
		std::vector<int64_t> rel;                  // the chunk's offsets, from 0
		Py_ssize_t s0 = 0, s1 = 0;
		std::thread th;
		int rc = GF2BV_OK;
		std::string err;
		bool running = false;
		~Chunk() { if (th.joinable()) th.join(); }      // (never left running: an exception on the way out must not meet a live thread)
	} ck[2];
	int rc = GF2BV_OK;
	std::string err;
	auto join = [&](Chunk &c) {
		if (!c.running) return;
		Py_BEGIN_ALLOW_THREADS
		c.th.join();
		Py_END_ALLOW_THREADS
		c.running = false;
		c.st.drop();
		if (c.rc != GF2BV_OK && rc == GF2BV_OK) { rc = c.rc; err = c.err; }
	};
	bool pyerr = false;
	int which = 0;
	for (Py_ssize_t s0 = 0; s0 < nsys && rc == GF2BV_OK; s0 += chunk_sys, which ^= 1) {
		Chunk &c = ck[which];
		join(c);                                   // (the chunk before the previous one: its buffer is free again)
		if (rc != GF2BV_OK) break;
		c.s0 = s0; c.s1 = std::min<Py_ssize_t>(nsys, s0 + chunk_sys);
		const size_t r0 = (size_t)(c.s0 * rows), r1 = (size_t)(c.s1 * rows);
		if (!dg.gather_rows(r0, r1, chunk_bytes + chunk_bytes / 2, c.st)) { pyerr = true; break; }
		try { c.rel.assign(dg.off.begin() + (ptrdiff_t)r0, dg.off.begin() + (ptrdiff_t)r1 + 1); }
		catch (const std::bad_alloc &) { PyErr_NoMemory(); pyerr = true; break; }
		const int64_t base = c.rel[0];
		for (int64_t &o : c.rel) o -= base;
		// (the share -> device map of gf2bv_solve_batch_digits_multi: share k = systems floor(n k / shares) ..)
		const Py_ssize_t n = c.s1 - c.s0, nsh = std::min<Py_ssize_t>((Py_ssize_t)devs.size(), n);
		for (Py_ssize_t t = 0, share = 0; t < n; t++) {
			while (share + 1 < nsh && n * (share + 1) / nsh <= t) share++;
			sysdev[(size_t)(c.s0 + t)] = devs[(size_t)share];
		}
		join(ck[which ^ 1]);                       // one chunk on the device(s) at a time
		if (rc != GF2BV_OK) break;
		c.rc = GF2BV_OK; c.err.clear();
		c.running = true;
		try {
			c.th = std::thread([&c, &res, &devs, rows, cols, mode]() {
				c.rc = gf2bv_solve_batch_digits_multi(c.st.digits, c.rel.data(), PyLong_SHIFT, c.s1 - c.s0, rows, cols, (int)mode,
				                                      devs.data(), (int)devs.size(), res.data() + c.s0);
				if (c.rc != GF2BV_OK) c.err = gf2bv_last_error();      // (the message is per thread)
			});
		} catch (const std::system_error &) {
			c.running = false;
			rc = GF2BV_ERR_HIP; err = "could not start the solve thread";
		}
	}
	join(ck[0]); join(ck[1]);
	if (pyerr || rc != GF2BV_OK) {
		for (gf2bv_result *r : res) if (r) gf2bv_result_free(r);
		if (!pyerr)
			PyErr_Format(rc == GF2BV_ERR_ARG ? PyExc_ValueError : PyExc_RuntimeError,
			             "gf2bv_amd: HIP solve failed (%d): %s", rc, err.c_str());
		return nullptr;
	}
	PyObject *out = PyList_New(nsys);
	Py_ssize_t done = 0;
	for (; out && done < nsys; done++) {
		PyObject *item = result_to_py(res[done], mode, sysdev[(size_t)done]);      // frees res[done]
		if (!item) { done++; Py_CLEAR(out); break; }
		PyList_SET_ITEM(out, done, item);
	}
	for (Py_ssize_t t = done; t < nsys; t++) gf2bv_result_free(res[t]);
	return out;
}

// to_bits(n, a): tuple of n bools, LSB first, zero-extended (_internal.c:504-531)
PyObject *py_to_bits(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 2) { PyErr_SetString(PyExc_TypeError, "to_bits requires 2 arguments"); return nullptr; }
	Py_ssize_t n = PyLong_AsSsize_t(args[0]);
	if (n < 0) {
		if (n == -1 && PyErr_Occurred()) return nullptr;
		PyErr_SetString(PyExc_ValueError, "n must be non-negative");
		return nullptr;
	}
	if (!PyLong_Check(args[1])) { PyErr_SetString(PyExc_TypeError, "a must be an integer"); return nullptr; }
	PyLongObject *a = (PyLongObject *)args[1];
	PyObject *t = PyTuple_New(n);
	if (!t) return nullptr;
	for (Py_ssize_t i = 0; i < n; i++) {
		PyObject *b = long_bit(a, i) ? Py_True : Py_False;
		Py_INCREF(b);
		PyTuple_SET_ITEM(t, i, b);
	}
	return t;
}

// xor_tuple(a, b): element-wise a[i] ^ b[i] (_internal.c:606-639)
PyObject *py_xor_tuple(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 2) { PyErr_SetString(PyExc_TypeError, "xor_tuple requires 2 arguments"); return nullptr; }
	PyObject *a = args[0], *b = args[1];
	if (!PyTuple_Check(a) || !PyTuple_Check(b)) {
		PyErr_SetString(PyExc_TypeError, "a and b must be tuples");
		return nullptr;
	}
	const Py_ssize_t n = PyTuple_GET_SIZE(a);
	if (n != PyTuple_GET_SIZE(b)) {
		PyErr_SetString(PyExc_ValueError, "The length of a and b is not equal");
		return nullptr;
	}
	PyObject *t = PyTuple_New(n);
	if (!t) return nullptr;
	for (Py_ssize_t i = 0; i < n; i++) {
		PyObject *x = PyNumber_Xor(PyTuple_GET_ITEM(a, i), PyTuple_GET_ITEM(b, i));
		if (!x) {
			Py_DECREF(t);
			PyErr_SetString(PyExc_TypeError, "Failed to compute xor, list items must be integers");
			return nullptr;
		}
		PyTuple_SET_ITEM(t, i, x);
	}
	return t;
}

// tuple_where(cond, a, b): np.where-like; OVERWRITES cond in place and returns it; a / b may be
// scalars (_internal.c:641-676)
PyObject *py_tuple_where(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 3) { PyErr_SetString(PyExc_TypeError, "tuple_where requires 3 arguments"); return nullptr; }
	PyObject *cond = args[0], *a = args[1], *b = args[2];
	if (!PyTuple_Check(cond)) { PyErr_SetString(PyExc_TypeError, "cond must be a list"); return nullptr; }
	const bool at = PyTuple_Check(a), bt = PyTuple_Check(b);
	const Py_ssize_t n = PyTuple_GET_SIZE(cond);
	if (at && PyTuple_GET_SIZE(a) != n) {
		PyErr_SetString(PyExc_ValueError, "The length of a and cond is not equal");
		return nullptr;
	}
	if (bt && PyTuple_GET_SIZE(b) != n) {
		PyErr_SetString(PyExc_ValueError, "The length of b and cond is not equal");
		return nullptr;
	}
	for (Py_ssize_t i = 0; i < n; i++) {
		PyObject *c = PyTuple_GET_ITEM(cond, i);
		int truth = PyObject_IsTrue(c);
		if (truth < 0) return nullptr;
		PyObject *pick = truth ? (at ? PyTuple_GET_ITEM(a, i) : a) : (bt ? PyTuple_GET_ITEM(b, i) : b);
		Py_INCREF(pick);
		PyTuple_SET_ITEM(cond, i, pick);
		Py_DECREF(c);
	}
	Py_INCREF(cond);
	return cond;
}

// mul_bit_quad(n, a, b, v, basis): OR basis[1+n+i(i-1)/2+j] into v for every j<i with
// a_i b_j ^ a_j b_i = 1 (_internal.c:538-604).  Caller-side helper of QuadraticSystem.
PyObject *py_mul_bit_quad(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 5) { PyErr_SetString(PyExc_TypeError, "mul_bit_quad requires 5 arguments"); return nullptr; }
	Py_ssize_t n = PyLong_AsSsize_t(args[0]);
	if (n <= 0) {
		if (n == -1 && PyErr_Occurred()) return nullptr;
		PyErr_SetString(PyExc_ValueError, "n must be positive");
		return nullptr;
	}
	if (!PyLong_Check(args[1]) || !PyLong_Check(args[2]) || !PyLong_Check(args[3])) {
		PyErr_SetString(PyExc_TypeError, "a and b and v must be integers");
		return nullptr;
	}
	PyObject *basis = args[4];
	if (!PyList_Check(basis)) { PyErr_SetString(PyExc_TypeError, "basis must be a list"); return nullptr; }
	if (PyList_GET_SIZE(basis) != 1 + n + n * (n - 1) / 2) {
		PyErr_SetString(PyExc_ValueError, "The length of basis is not correct");
		return nullptr;
	}
	PyLongObject *a = (PyLongObject *)args[1], *b = (PyLongObject *)args[2];
	std::vector<char> ab((size_t)n), bb((size_t)n);
	for (Py_ssize_t i = 0; i < n; i++) { ab[i] = long_bit(a, i); bb[i] = long_bit(b, i); }
	PyObject *v = args[3];
	Py_INCREF(v);
	Py_ssize_t mi = 1 + n;
	for (Py_ssize_t i = 0; i < n; i++)
		for (Py_ssize_t j = 0; j < i; j++, mi++)
			if ((ab[i] & bb[j]) ^ (ab[j] & bb[i])) {
				PyObject *nv = PyNumber_Or(v, PyList_GET_ITEM(basis, mi));
				Py_DECREF(v);
				if (!nv) {
					PyErr_SetString(PyExc_TypeError, "Failed to compute or, list items must be integers");
					return nullptr;
				}
				v = nv;
			}
	return v;
}

// eqs_to_sage_mat_helper(eqs, cols) -> (png_bytes, affine_list)      (_internal.c:678-765)
// The reference renders the coefficient matrix as an image -- pixel (x = column, y = row) BLACK where the bit is set -- through
// libgd (dlopen at first use) and returns gdImagePngPtrEx(im, &size, 0): a palette PNG with colour 0 = black, 1 = white,
// compression level 0, which Sage's unpickle_matrix_mod2_dense_v2 reads back as entry = 1 - palette index.  Here the PNG is
// written directly (no libgd, no libpng, no zlib): signature, IHDR (bit depth 1, colour type 3 -- what libgd's writer picks
// for a two-colour palette), PLTE {000000, FFFFFF}, the scanlines (filter byte 0, pixels MSB first, padding bits white) as
// STORED deflate blocks (level 0) in IDAT chunks of at most 1 MiB, IEND.  Same pixels and palette as the reference's image;
// byte-for-byte equality with a particular libgd/libpng build is not claimed (chunking and zlib framing are the encoder's).
// affine[i] = bit 0 of eqs[i] as a bool; the reference leaves a NULL slot for an equation that is the int 0 (its bit loop
// never runs, _internal.c:741-752) -- get_eqs never passes one -- here that entry is False.
namespace png {
uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(uint32_t c, const unsigned char *p, size_t n)
{
	if (!crc_ready) {
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t v = i;
			for (int k = 0; k < 8; k++) v = (v & 1) ? 0xEDB88320u ^ (v >> 1) : v >> 1;
			crc_table[i] = v;
		}
		crc_ready = true;
	}
	c = ~c;
	for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
	return ~c;
}
void be32(std::vector<unsigned char> &o, uint32_t v) { for (int s = 24; s >= 0; s -= 8) o.push_back((unsigned char)(v >> s)); }
void chunk(std::vector<unsigned char> &o, const char *type, const unsigned char *data, size_t n)
{
	be32(o, (uint32_t)n);
	const size_t at = o.size();
	o.insert(o.end(), type, type + 4);
	o.insert(o.end(), data, data + n);
	be32(o, crc32(0, o.data() + at, n + 4));
}
}  // namespace png

PyObject *py_sage_helper(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 2) { PyErr_SetString(PyExc_TypeError, "eqs_to_sage_mat_helper requires 2 arguments"); return nullptr; }
	PyObject *list = args[0];
	if (!PyList_Check(list)) { PyErr_SetString(PyExc_TypeError, "The first argument equations must be a list"); return nullptr; }
	const Py_ssize_t cols = PyLong_AsSsize_t(args[1]);
	if (cols <= 0) {
		if (cols == -1 && PyErr_Occurred()) return nullptr;
		PyErr_SetString(PyExc_ValueError, "Number of columns must be positive");
		return nullptr;
	}
	const Py_ssize_t rows = PyList_GET_SIZE(list);
	if (cols > 0x7fffffff || rows > 0x7fffffff) { PyErr_SetString(PyExc_ValueError, "image dimensions exceed PNG's 2^31 - 1"); return nullptr; }
	for (Py_ssize_t i = 0; i < rows; i++)
		if (!PyLong_Check(PyList_GET_ITEM(list, i))) { PyErr_SetString(PyExc_TypeError, "All elements in the equations list must be integers"); return nullptr; }
	PyObject *affine = PyList_New(rows);
	if (!affine) return nullptr;
	try {
		const size_t rb = (size_t)(cols + 7) / 8, line = rb + 1;
		const size_t words = (size_t)(cols + 63) / 64;
		// the raw image (filter byte + packed pixels per row)
		std::vector<unsigned char> raw(line * (size_t)rows);
		std::vector<uint64_t> bits(words + 1);
		static unsigned char rev[256];
		static bool rev_ready = false;
		if (!rev_ready) { for (int v = 0; v < 256; v++) { int r = 0; for (int k = 0; k < 8; k++) if (v >> k & 1) r |= 0x80 >> k; rev[v] = (unsigned char)r; } rev_ready = true; }
		for (Py_ssize_t i = 0; i < rows; i++) {
			PyLongObject *v = (PyLongObject *)PyList_GET_ITEM(list, i);
			const Py_ssize_t nd = GF2_DIGIT_COUNT(v);
			PyObject *aff = (nd > 0 && (GF2_DIGITS(v)[0] & 1)) ? Py_True : Py_False;
			Py_INCREF(aff);
			PyList_SET_ITEM(affine, i, aff);
			// bits[c] = bit c + 1 of |v| for c < cols (the sign is ignored, bits above cols are ignored: _internal.c:41-59)
			std::fill(bits.begin(), bits.end(), 0);
			for (Py_ssize_t d = 0; d < nd; d++) {
				uint64_t val = GF2_DIGITS(v)[d];
				Py_ssize_t at = d * PyLong_SHIFT - 1;
				if (d == 0) { val >>= 1; at = 0; }
				if (at >= cols) break;
				bits[(size_t)at >> 6] |= val << (at & 63);
				if ((at & 63) + PyLong_SHIFT > 64) bits[((size_t)at >> 6) + 1] |= val >> (64 - (at & 63));
			}
			if (cols & 63) bits[words - 1] &= (~0ull >> (64 - (cols & 63)));
			unsigned char *dst = raw.data() + line * (size_t)i;
			dst[0] = 0;                                       // filter type None
			for (size_t k = 0; k < rb; k++) dst[1 + k] = rev[(unsigned char)~(bits[k >> 3] >> (8 * (k & 7)))];      // set bit -> index 0 (black)
		}
		// zlib stream of stored blocks
		std::vector<unsigned char> z;
		z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
		z.push_back(0x78); z.push_back(0x01);
		uint32_t a1 = 1, a2 = 0;
		size_t pos = 0;
		do {
			const size_t n = std::min<size_t>(65535, raw.size() - pos);
			z.push_back(pos + n == raw.size() ? 1 : 0);
			z.push_back((unsigned char)(n & 0xff)); z.push_back((unsigned char)(n >> 8));
			z.push_back((unsigned char)(~n & 0xff)); z.push_back((unsigned char)((~n >> 8) & 0xff));
			z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
			for (size_t k = 0; k < n;) {                      // adler32, reduced every 5552 bytes
				const size_t m = std::min<size_t>(5552, n - k);
				for (size_t e = 0; e < m; e++) { a1 += raw[pos + k + e]; a2 += a1; }
				a1 %= 65521; a2 %= 65521; k += m;
			}
			pos += n;
		} while (pos < raw.size());
		png::be32(z, (a2 << 16) | a1);
		std::vector<unsigned char>().swap(raw);
		std::vector<unsigned char> out;
		out.reserve(z.size() + z.size() / (1 << 20) * 12 + 128);
		static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
		out.insert(out.end(), sig, sig + 8);
		std::vector<unsigned char> ihdr;
		png::be32(ihdr, (uint32_t)cols); png::be32(ihdr, (uint32_t)rows);
		ihdr.push_back(1); ihdr.push_back(3); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
		png::chunk(out, "IHDR", ihdr.data(), ihdr.size());
		static const unsigned char plte[6] = { 0, 0, 0, 255, 255, 255 };
		png::chunk(out, "PLTE", plte, 6);
		for (size_t at = 0; at < z.size(); at += (size_t)1 << 20)
			png::chunk(out, "IDAT", z.data() + at, std::min<size_t>((size_t)1 << 20, z.size() - at));
		png::chunk(out, "IEND", nullptr, 0);
		PyObject *bytes = PyBytes_FromStringAndSize((const char *)out.data(), (Py_ssize_t)out.size());
		if (!bytes) { Py_DECREF(affine); return nullptr; }
		PyObject *ret = PyTuple_Pack(2, bytes, affine);
		Py_DECREF(bytes); Py_DECREF(affine);
		return ret;
	} catch (const std::bad_alloc &) {
		Py_DECREF(affine);
		return PyErr_NoMemory();
	}
}

// _space_from_ints(cols, origin, basis): build an AffineSpace from host integers.  Not part of the
// reference surface; lets the CPU test-suite exercise get()/iteration order without a GPU.
PyObject *py_space_from_ints(PyObject *, PyObject *const *args, Py_ssize_t nargs)
{
	if (nargs != 3 || !PyLong_Check(args[1]) || !PyTuple_Check(args[2])) {
		PyErr_SetString(PyExc_TypeError, "_space_from_ints(cols, origin, basis_tuple)");
		return nullptr;
	}
	Py_ssize_t cols = PyLong_AsSsize_t(args[0]);
	if (cols <= 0) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "cols must be positive"); return nullptr; }
	const int64_t words = (cols + 63) / 64;
	const int64_t dim = PyTuple_GET_SIZE(args[2]);
	AffineSpaceObject *sp = (AffineSpaceObject *)AffineSpace_Type->tp_alloc(AffineSpace_Type, 0);
	if (!sp) return nullptr;
	sp->dim = dim;
	sp->words = words;
	sp->device = -1;                 // host-built: walked on the host (no GPU needed for the host-logic tests)
	sp->origin = (uint64_t *)calloc((size_t)words, sizeof(uint64_t));
	sp->basis = (uint64_t *)calloc((size_t)((dim * words) > 0 ? dim * words : 1), sizeof(uint64_t));
	if (!sp->origin || !sp->basis) { Py_DECREF(sp); return PyErr_NoMemory(); }
	for (int64_t k = -1; k < dim; k++) {
		PyObject *v = k < 0 ? args[1] : PyTuple_GET_ITEM(args[2], k);
		if (!PyLong_Check(v)) { Py_DECREF(sp); PyErr_SetString(PyExc_TypeError, "integers expected"); return nullptr; }
		uint64_t *dst = k < 0 ? sp->origin : sp->basis + k * words;
		for (Py_ssize_t i = 0; i < cols; i++)
			if (long_bit((PyLongObject *)v, i)) dst[i >> 6] |= (uint64_t)1 << (i & 63);
	}
	return (PyObject *)sp;
}

PyObject *py_device_count(PyObject *, PyObject *) { return PyLong_FromLong(gf2bv_device_count()); }
PyObject *py_get_default_device(PyObject *, PyObject *) { return PyLong_FromLong(default_device()); }
PyObject *py_set_default_device(PyObject *, PyObject *arg)
{
	int d;
	if (arg == Py_None) { PyErr_SetString(PyExc_TypeError, "device must be an int"); return nullptr; }
	if (!parse_device(arg, &d)) return nullptr;
	g_default_device = d;
	Py_RETURN_NONE;
}

// content ids of the two binaries (gf2bv_amd/build.py compiles them in; the marker lets build.py read the shim's without loading it)
#ifndef GF2BV_SHIM_ID
#define GF2BV_SHIM_ID "unknown"
#endif
const char g_shim_id[] = "GF2BV_SHIM_ID=" GF2BV_SHIM_ID;
PyObject *py_build_id(PyObject *, PyObject *) { return Py_BuildValue("{s:s,s:s}", "hip", gf2bv_build_id(), "shim", g_shim_id + 14); }

#define FAST(fn) (PyCFunction)(void (*)(void))(fn)
PyMethodDef module_methods[] = {
	{"m4ri_solve", FAST(py_m4ri_solve), METH_FASTCALL,
	 "m4ri_solve(equations, cols, mode, device=None)\n--\n\nSolve the linear system on the MI355X; None when inconsistent."},
	{"m4ri_solve_packed", FAST(py_m4ri_solve_packed), METH_FASTCALL,
	 "m4ri_solve_packed(buffer, rows, words, cols, mode, device=None)\n--\n\nm4ri_solve on equations already packed as rows x words 64-bit words (equation-int bit order)."},
	{"m4ri_solve_many", FAST(py_m4ri_solve_many), METH_FASTCALL,
	 "m4ri_solve_many(systems, cols, mode, devices=None)\n--\n\nSolve a list of same-shape systems in one batched call, sharded over the given GPUs (None: the default device, \"all\": every visible one); list of m4ri_solve results."},
	{"to_bits", FAST(py_to_bits), METH_FASTCALL, "to_bits(n, a)\n--\n\nLow n bits of a, LSB first."},
	{"mul_bit_quad", FAST(py_mul_bit_quad), METH_FASTCALL, "mul_bit_quad(n, a, b, v, basis)\n--\n\n"},
	{"xor_tuple", FAST(py_xor_tuple), METH_FASTCALL, "xor_tuple(a, b)\n--\n\nElement-wise xor."},
	{"tuple_where", FAST(py_tuple_where), METH_FASTCALL, "tuple_where(cond, a, b)\n--\n\nIn-place select."},
	{"eqs_to_sage_mat_helper", FAST(py_sage_helper), METH_FASTCALL,
	 "eqs_to_sage_mat_helper(equations, cols)\n--\n\n(png_bytes, affine): the coefficient matrix as a two-colour PNG (black = 1) for Sage's unpickle_matrix_mod2_dense_v2, and the affine bits."},
	{"_space_from_ints", FAST(py_space_from_ints), METH_FASTCALL, "test hook: AffineSpace from ints"},
	{"device_count", py_device_count, METH_NOARGS, "number of visible HIP devices"},
	{"build_id", py_build_id, METH_NOARGS, "content hashes of the sources libgf2bv_hip.so and this module were compiled from"},
	{"get_default_device", py_get_default_device, METH_NOARGS, "the device solves run on when no `device` argument is given"},
	{"set_default_device", py_set_default_device, METH_O, "set_default_device(d)\n--\n\nselect the GPU for solves without a `device` argument"},
	{nullptr, nullptr, 0, nullptr}};

PyModuleDef module_def = {PyModuleDef_HEAD_INIT, "_internal", "gf2bv_amd native boundary (HIP solver)", -1,
                          module_methods, nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__internal(void)
{
	PyObject *m = PyModule_Create(&module_def);
	if (!m) return nullptr;
	AffineSpace_Type = (PyTypeObject *)PyType_FromSpec(&space_spec);
	SpaceIterGray_Type = (PyTypeObject *)PyType_FromSpec(&gray_spec);
	SpaceIterSlow_Type = (PyTypeObject *)PyType_FromSpec(&slow_spec);
	if (!AffineSpace_Type || !SpaceIterGray_Type || !SpaceIterSlow_Type) { Py_DECREF(m); return nullptr; }
	Py_INCREF(AffineSpace_Type); Py_INCREF(SpaceIterGray_Type); Py_INCREF(SpaceIterSlow_Type);
	if (PyModule_AddObject(m, "AffineSpace", (PyObject *)AffineSpace_Type) < 0 ||
	    PyModule_AddObject(m, "AffineSpaceIterator", (PyObject *)SpaceIterGray_Type) < 0 ||
	    PyModule_AddObject(m, "AffineSpaceIteratorSlow", (PyObject *)SpaceIterSlow_Type) < 0) {
		Py_DECREF(m);
		return nullptr;
	}
	return m;
}
