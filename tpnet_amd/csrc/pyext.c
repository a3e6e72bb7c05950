/* _tpnet_fast -- CPython extension in front of the C ABI (include/tpnet_hip.h) for the calls the reference's training loop
 * makes once per batch (train_link_prediction.py:325-373: two get_pair_wise_feature, one update) and for the stream call.
 * BASELINE.json's north_star names "a tiny PyTorch-ROCm C++/HIP extension that keeps the TPNet.forward / update signatures":
 * the signatures live in tpnet_amd/random_projection.py; this file is the crossing itself -- METH_FASTCALL entry points
 * that take the host numpy arrays through the buffer protocol and plain integers for device pointers, so that a call costs a
 * fraction of a microsecond of marshalling instead of ctypes' 4-6 us for a dozen arguments.  No logic lives here: every function
 * forwards to exactly one entry point of libtpnet_hip.so (linked, rpath $ORIGIN).  Built by tpnet_amd/csrc/Makefile with gcc. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#include "../../include/tpnet_hip.h"

static int as_u64(PyObject* o, uint64_t* out) {
    if (o == Py_None) { *out = 0; return 1; }
    const unsigned long long v = PyLong_AsUnsignedLongLong(o);
    if (v == (unsigned long long)-1 && PyErr_Occurred()) return 0;
    *out = (uint64_t)v;
    return 1;
}
static int as_i64(PyObject* o, int64_t* out) {
    const long long v = PyLong_AsLongLong(o);
    if (v == -1 && PyErr_Occurred()) return 0;
    *out = (int64_t)v;
    return 1;
}
static int as_f64(PyObject* o, double* out) {
    const double v = PyFloat_AsDouble(o);
    if (v == -1.0 && PyErr_Occurred()) return 0;
    *out = v;
    return 1;
}

/* a one-dimensional C-contiguous array of 8-byte items (int64 ids, float64 times) through the buffer protocol */
static int get_vec8(PyObject* o, Py_buffer* view, const char* what) {
    if (PyObject_GetBuffer(o, view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return 0;
    if (view->ndim != 1 || view->itemsize != 8) {
        PyBuffer_Release(view);
        PyErr_Format(PyExc_ValueError, "%s must be a one-dimensional contiguous array of 8-byte items", what);
        return 0;
    }
    return 1;
}

/* pair_feature(state, stage, u, v, now, lam, flags, mlp, out_gram, out, stream) -> status
 * tpnet_host_pair_feature: state / stage / mlp = addresses of the tpnet_state struct, the stage handle, a tpnet_mlp struct (0: none) */
static PyObject* f_pair_feature(PyObject* self, PyObject* const* a, Py_ssize_t n) {
    (void)self;
    if (n != 11) { PyErr_SetString(PyExc_TypeError, "pair_feature takes 11 arguments"); return NULL; }
    uint64_t st, stage, mlp, out_gram, out, stream;
    int64_t flags;
    double now, lam;
    if (!as_u64(a[0], &st) || !as_u64(a[1], &stage) || !as_f64(a[4], &now) || !as_f64(a[5], &lam) || !as_i64(a[6], &flags) ||
        !as_u64(a[7], &mlp) || !as_u64(a[8], &out_gram) || !as_u64(a[9], &out) || !as_u64(a[10], &stream))
        return NULL;
    Py_buffer u, v;
    if (!get_vec8(a[2], &u, "src_node_ids")) return NULL;
    if (!get_vec8(a[3], &v, "dst_node_ids")) { PyBuffer_Release(&u); return NULL; }
    int rc;
    if (u.shape[0] != v.shape[0]) {
        rc = TPNET_ERR_BAD_ARG;
    } else {
        rc = tpnet_host_pair_feature((const tpnet_state*)(uintptr_t)st, (tpnet_stage*)(uintptr_t)stage, (const int64_t*)u.buf,
                                     (const int64_t*)v.buf, (int64_t)u.shape[0], now, lam, (uint32_t)flags,
                                     (const tpnet_mlp*)(uintptr_t)mlp, (float*)(uintptr_t)out_gram, (float*)(uintptr_t)out,
                                     (void*)(uintptr_t)stream);
    }
    PyBuffer_Release(&u);
    PyBuffer_Release(&v);
    return PyLong_FromLong(rc);
}

/* update(state, stage, src, dst, t, now, lam, launch_id, flags, ws, ws_bytes, stream) -> status    (tpnet_host_update) */
static PyObject* f_update(PyObject* self, PyObject* const* a, Py_ssize_t n) {
    (void)self;
    if (n != 12) { PyErr_SetString(PyExc_TypeError, "update takes 12 arguments"); return NULL; }
    uint64_t st, stage, ws, ws_bytes, stream;
    int64_t lid, flags;
    double now, lam;
    if (!as_u64(a[0], &st) || !as_u64(a[1], &stage) || !as_f64(a[5], &now) || !as_f64(a[6], &lam) || !as_i64(a[7], &lid) ||
        !as_i64(a[8], &flags) || !as_u64(a[9], &ws) || !as_u64(a[10], &ws_bytes) || !as_u64(a[11], &stream))
        return NULL;
    Py_buffer s, d, t;
    if (!get_vec8(a[2], &s, "src_node_ids")) return NULL;
    if (!get_vec8(a[3], &d, "dst_node_ids")) { PyBuffer_Release(&s); return NULL; }
    if (!get_vec8(a[4], &t, "node_interact_times")) { PyBuffer_Release(&s); PyBuffer_Release(&d); return NULL; }
    int rc;
    if (s.shape[0] != d.shape[0] || s.shape[0] != t.shape[0]) {
        rc = TPNET_ERR_BAD_ARG;
    } else {
        rc = tpnet_host_update((const tpnet_state*)(uintptr_t)st, (tpnet_stage*)(uintptr_t)stage, (const int64_t*)s.buf,
                               (const int64_t*)d.buf, (const double*)t.buf, (int64_t)s.shape[0], now, lam, (uint32_t)lid,
                               (uint32_t)flags, (void*)(uintptr_t)ws, (size_t)ws_bytes, (void*)(uintptr_t)stream);
    }
    PyBuffer_Release(&s);
    PyBuffer_Release(&d);
    PyBuffer_Release(&t);
    return PyLong_FromLong(rc);
}

/* run_stream(state, src, dst, neg, t, E, batch, now, lam, launch_id, flags, out_pos, out_neg, ws, ws_bytes, want_t_end, stream,
 *            tag) -> (status, t_end)      (tpnet_run_stream_tagged; device pointers and the tag's address as integers) */
static PyObject* f_run_stream(PyObject* self, PyObject* const* a, Py_ssize_t n) {
    (void)self;
    if (n != 18) { PyErr_SetString(PyExc_TypeError, "run_stream takes 18 arguments"); return NULL; }
    uint64_t st, src, dst, neg, t, out_pos, out_neg, ws, ws_bytes, stream, tag;
    int64_t E, batch, lid, flags, want_t;
    double now, lam;
    if (!as_u64(a[0], &st) || !as_u64(a[1], &src) || !as_u64(a[2], &dst) || !as_u64(a[3], &neg) || !as_u64(a[4], &t) ||
        !as_i64(a[5], &E) || !as_i64(a[6], &batch) || !as_f64(a[7], &now) || !as_f64(a[8], &lam) || !as_i64(a[9], &lid) ||
        !as_i64(a[10], &flags) || !as_u64(a[11], &out_pos) || !as_u64(a[12], &out_neg) || !as_u64(a[13], &ws) ||
        !as_u64(a[14], &ws_bytes) || !as_i64(a[15], &want_t) || !as_u64(a[16], &stream) || !as_u64(a[17], &tag))
        return NULL;
    double t_end = 0.0;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = tpnet_run_stream_tagged((const tpnet_state*)(uintptr_t)st, (const int64_t*)(uintptr_t)src, (const int64_t*)(uintptr_t)dst,
                                 (const int64_t*)(uintptr_t)neg, (const double*)(uintptr_t)t, E, batch, now, lam, (uint32_t)lid,
                                 (uint32_t)flags, (float*)(uintptr_t)out_pos, (float*)(uintptr_t)out_neg, (void*)(uintptr_t)ws,
                                 (size_t)ws_bytes, want_t ? &t_end : NULL, (void*)(uintptr_t)stream,
                                 (tpnet_plan_tag*)(uintptr_t)tag);
    Py_END_ALLOW_THREADS
    return Py_BuildValue("(id)", rc, t_end);
}

static PyObject* f_abi_version(PyObject* self, PyObject* noargs) {
    (void)self; (void)noargs;
    return PyLong_FromLong(tpnet_abi_version());
}

static PyMethodDef methods[] = {
    {"pair_feature", (PyCFunction)(void (*)(void))f_pair_feature, METH_FASTCALL, "tpnet_host_pair_feature"},
    {"update", (PyCFunction)(void (*)(void))f_update, METH_FASTCALL, "tpnet_host_update"},
    {"run_stream", (PyCFunction)(void (*)(void))f_run_stream, METH_FASTCALL, "tpnet_run_stream_tagged"},
    {"abi_version", f_abi_version, METH_NOARGS, "tpnet_abi_version of the library this module is linked to"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_tpnet_fast",
                                    "fast crossing to libtpnet_hip.so for the per-batch calls", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__tpnet_fast(void) { return PyModule_Create(&moddef); }
