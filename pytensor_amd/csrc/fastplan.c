/* fastplan.c — the per-call host path of a frozen plan as ONE native call (CPython extension `_fastplan`).
 *
 * What the reference does per `Function.__call__` on its fastest runtime is a single C call into the CVM
 * (pytensor/link/c/c_code/lazylinker_c.c:749 `CLazyLinker_call`; the Python side is compile/executor.py:651-744).
 * The hip linker's replay path was ~15 us of Python per evaluation of config #4 (profiles/r4b_c4_timeline: 24 us
 * between the end of one evaluation's last kernel and the start of the next one's first copy): five staged inputs
 * checked and copied through NumPy, four residents checked through ctypes, a 4-argument ctypes call, six result
 * arrays built one by one.  `FastPlan.__call__(inputs)` does the same work natively:
 *   1. every staged input: exact ndarray, same dtype / shape as captured, C-contiguous -> memcpy into the pinned
 *      staging block;
 *   2. every resident input: the very object that was uploaded, its write-protection slot still clean
 *      (pthip_guard_clean: dirty flag + the two ragged ends);
 *   3. pthip_plan_replay4 (upload, launches, completion poll) with the GIL released;
 *   4. device status word; fresh result arrays (a copy out of the pinned result block each, 0-d -> NumPy scalar
 *      where the graph output is a ScalarType).
 * Anything it cannot vouch for (another signature, a dirty resident, a non-contiguous argument) returns None
 * BEFORE anything is launched and the Python path (plan.py FrozenPlan.__call__) handles the call; a non-zero
 * device status word returns the int (the results are not built).  No arithmetic happens here: every number
 * still comes out of the HIP kernels the plan replays.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <string.h>

typedef int (*replay4_fn)(const void* desc, void* host_out, volatile int* done, int sync);
typedef int (*guard_clean_fn)(int slot);

#define FP_MAXDIM 8

typedef struct {
  int pos;
  char* dst;
  Py_ssize_t nbytes;
  int typenum, ndim;
  npy_intp shape[FP_MAXDIM];
} Staged;

typedef struct {
  int pos;
  PyObject* obj; /* owned: the address cannot be recycled for another array while this plan compares against it */
  int slot;      /* guard slot, or -1: nothing to check (coherence mode "trust") */
} Resident;

typedef struct {
  const char* src; /* in the pinned result block; NULL: a host-known constant output */
  PyObject* konst; /* owned */
  int typenum, ndim, scalar;
  Py_ssize_t nbytes;
  npy_intp shape[FP_MAXDIM];
} Output;

typedef struct {
  PyObject_HEAD
  int n_inputs, n_staged, n_res, n_out;
  Staged* staged;
  Resident* res;
  Output* out;
  replay4_fn replay;
  guard_clean_fn gclean;
  const void* desc;
  volatile int* done;
  volatile int* status;
  int sync_mode;
  /* the executable's resident generation (HipExecutable._res_gen): bumped by invalidate_resident, by every
   * re-upload into a resident's device buffer (another plan of the same executable, an eager call, set_value)
   * and by every re-watch.  A plan vouches for its residents only while the counter still has the value it had
   * when this object was built from a validated Python-path call (in mode "trust" there is no guard
   * slot to look at, and a released slot number can be reused for another array). */
  const volatile unsigned long long* gen;
  unsigned long long gen_built;
  unsigned long long calls, misses;
} FastPlan;

static void FastPlan_dealloc(FastPlan* self) {
  if (self->out)
    for (int k = 0; k < self->n_out; k++) Py_XDECREF(self->out[k].konst);
  if (self->res)
    for (int k = 0; k < self->n_res; k++) Py_XDECREF(self->res[k].obj);
  PyMem_Free(self->staged);
  PyMem_Free(self->res);
  PyMem_Free(self->out);
  Py_TYPE(self)->tp_free((PyObject*)self);
}

static int fill_shape(PyObject* seq, npy_intp* shape, int* ndim) {
  PyObject* t = PySequence_Tuple(seq);
  if (!t) return -1;
  Py_ssize_t n = PyTuple_GET_SIZE(t);
  if (n > FP_MAXDIM) { Py_DECREF(t); PyErr_SetString(PyExc_ValueError, "fastplan: more than 8 dimensions"); return -1; }
  for (Py_ssize_t i = 0; i < n; i++) {
    shape[i] = (npy_intp)PyLong_AsSsize_t(PyTuple_GET_ITEM(t, i));
    if (shape[i] == -1 && PyErr_Occurred()) { Py_DECREF(t); return -1; }
  }
  *ndim = (int)n;
  Py_DECREF(t);
  return 0;
}

/* FastPlan(n_inputs, staged, residents, outputs, replay_addr, guard_clean_addr, desc_addr, done_addr, status_addr, sync_mode,
 *          gen_addr, gen_value)
 *   staged    [(pos, dst_addr, dtype, shape)]
 *   residents [(pos, obj, slot)]
 *   outputs   [(src_addr | None, const | None, dtype, shape, scalar)] */
static int FastPlan_init(FastPlan* self, PyObject* args, PyObject* kwds) {
  PyObject *staged, *res, *outs;
  unsigned long long replay, gclean, desc, done, status, gen, gen_value;
  int n_inputs, sync_mode;
  if (!PyArg_ParseTuple(args, "iOOOKKKKKiKK", &n_inputs, &staged, &res, &outs, &replay, &gclean, &desc, &done, &status, &sync_mode, &gen, &gen_value)) return -1;
  if (!gen) { PyErr_SetString(PyExc_ValueError, "fastplan: the resident generation counter is required"); return -1; }
  self->gen = (const volatile unsigned long long*)(uintptr_t)gen;
  self->gen_built = gen_value;
  self->n_inputs = n_inputs;
  self->replay = (replay4_fn)(uintptr_t)replay;
  self->gclean = (guard_clean_fn)(uintptr_t)gclean;
  self->desc = (const void*)(uintptr_t)desc;
  self->done = (volatile int*)(uintptr_t)done;
  self->status = (volatile int*)(uintptr_t)status;
  self->sync_mode = sync_mode;
  self->n_staged = (int)PyList_Size(staged);
  self->n_res = (int)PyList_Size(res);
  self->n_out = (int)PyList_Size(outs);
  if (PyErr_Occurred()) return -1;
  self->staged = PyMem_Calloc(self->n_staged ? self->n_staged : 1, sizeof(Staged));
  self->res = PyMem_Calloc(self->n_res ? self->n_res : 1, sizeof(Resident));
  self->out = PyMem_Calloc(self->n_out ? self->n_out : 1, sizeof(Output));
  if (!self->staged || !self->res || !self->out) { PyErr_NoMemory(); return -1; }
  for (int k = 0; k < self->n_staged; k++) {
    PyObject *dt, *shape;
    unsigned long long dst;
    Staged* s = &self->staged[k];
    if (!PyArg_ParseTuple(PyList_GET_ITEM(staged, k), "iKOO", &s->pos, &dst, &dt, &shape)) return -1;
    PyArray_Descr* d = NULL;
    if (!PyArray_DescrConverter(dt, &d)) return -1;
    s->typenum = d->type_num;
    s->nbytes = (Py_ssize_t)PyDataType_ELSIZE(d);
    Py_DECREF(d);
    s->dst = (char*)(uintptr_t)dst;
    if (fill_shape(shape, s->shape, &s->ndim)) return -1;
    for (int i = 0; i < s->ndim; i++) s->nbytes *= s->shape[i];
    if (s->pos < 0 || s->pos >= n_inputs) { PyErr_SetString(PyExc_ValueError, "fastplan: staged position out of range"); return -1; }
  }
  for (int k = 0; k < self->n_res; k++) {
    Resident* r = &self->res[k];
    PyObject* obj;
    if (!PyArg_ParseTuple(PyList_GET_ITEM(res, k), "iOi", &r->pos, &obj, &r->slot)) return -1;
    Py_INCREF(obj);
    r->obj = obj;
    if (r->pos < 0 || r->pos >= n_inputs) { PyErr_SetString(PyExc_ValueError, "fastplan: resident position out of range"); return -1; }
  }
  for (int k = 0; k < self->n_out; k++) {
    PyObject *src, *konst, *dt, *shape;
    Output* o = &self->out[k];
    if (!PyArg_ParseTuple(PyList_GET_ITEM(outs, k), "OOOOi", &src, &konst, &dt, &shape, &o->scalar)) return -1;
    PyArray_Descr* d = NULL;
    if (!PyArray_DescrConverter(dt, &d)) return -1;
    o->typenum = d->type_num;
    o->nbytes = (Py_ssize_t)PyDataType_ELSIZE(d);
    Py_DECREF(d);
    if (fill_shape(shape, o->shape, &o->ndim)) return -1;
    for (int i = 0; i < o->ndim; i++) o->nbytes *= o->shape[i];
    if (src == Py_None) {
      if (!PyArray_Check(konst)) { PyErr_SetString(PyExc_TypeError, "fastplan: a constant output must be an ndarray"); return -1; }
      Py_INCREF(konst);
      o->konst = konst;
    } else {
      o->src = (const char*)(uintptr_t)PyLong_AsUnsignedLongLong(src);
      if (PyErr_Occurred()) return -1;
    }
  }
  return 0;
}

static PyObject* FastPlan_call(FastPlan* self, PyObject* args, PyObject* kwds) {
  PyObject* inputs;
  if (kwds && PyDict_GET_SIZE(kwds)) { PyErr_SetString(PyExc_TypeError, "FastPlan takes no keyword arguments"); return NULL; }
  if (PyTuple_GET_SIZE(args) != 1) { PyErr_SetString(PyExc_TypeError, "FastPlan(inputs_tuple)"); return NULL; }
  inputs = PyTuple_GET_ITEM(args, 0);
  self->calls++;
  if (!PyTuple_CheckExact(inputs) || PyTuple_GET_SIZE(inputs) != self->n_inputs) goto miss;
  /* residents first: nothing may be touched before every check has passed */
  if (self->n_res && *self->gen != self->gen_built) goto miss;
  for (int k = 0; k < self->n_res; k++) {
    const Resident* r = &self->res[k];
    if (PyTuple_GET_ITEM(inputs, r->pos) != r->obj) goto miss;
    if (r->slot >= 0 && self->gclean(r->slot) != 1) goto miss;
  }
  for (int k = 0; k < self->n_staged; k++) {
    const Staged* s = &self->staged[k];
    PyObject* a = PyTuple_GET_ITEM(inputs, s->pos);
    if (!PyArray_CheckExact(a)) goto miss;
    PyArrayObject* arr = (PyArrayObject*)a;
    if (PyArray_TYPE(arr) != s->typenum || PyArray_NDIM(arr) != s->ndim || !PyArray_IS_C_CONTIGUOUS(arr) || !PyArray_ISNOTSWAPPED(arr)) goto miss;
    const npy_intp* sh = PyArray_DIMS(arr);
    for (int i = 0; i < s->ndim; i++)
      if (sh[i] != s->shape[i]) goto miss;
  }
  for (int k = 0; k < self->n_staged; k++) {
    const Staged* s = &self->staged[k];
    memcpy(s->dst, PyArray_DATA((PyArrayObject*)PyTuple_GET_ITEM(inputs, s->pos)), (size_t)s->nbytes);
  }
  int rc;
  Py_BEGIN_ALLOW_THREADS
  rc = self->replay(self->desc, NULL, self->done, self->sync_mode);
  Py_END_ALLOW_THREADS
  if (rc) return PyLong_FromLong(-(long)(rc > 0 ? rc : -rc)); /* a HIP error: the Python side raises it from pthip_last_error */
  if (self->status && *self->status) return PyLong_FromLong((long)*self->status);
  PyObject* res = PyTuple_New(self->n_out);
  if (!res) return NULL;
  for (int k = 0; k < self->n_out; k++) {
    const Output* o = &self->out[k];
    PyObject* v;
    if (o->konst) {
      v = (PyObject*)PyArray_NewCopy((PyArrayObject*)o->konst, NPY_CORDER);
    } else {
      v = PyArray_SimpleNew(o->ndim, (npy_intp*)o->shape, o->typenum);
      if (v && o->nbytes) memcpy(PyArray_DATA((PyArrayObject*)v), o->src, (size_t)o->nbytes);
    }
    if (!v) { Py_DECREF(res); return NULL; }
    if (o->scalar) v = PyArray_Return((PyArrayObject*)v); /* 0-d -> NumPy scalar (steals the reference) */
    PyTuple_SET_ITEM(res, k, v);
  }
  return res;
miss:
  self->misses++;
  Py_RETURN_NONE;
}

static PyObject* FastPlan_stats(FastPlan* self, PyObject* Py_UNUSED(ignored)) {
  return Py_BuildValue("{s:K,s:K}", "calls", self->calls, "misses", self->misses);
}

static PyMethodDef FastPlan_methods[] = {
    {"stats", (PyCFunction)FastPlan_stats, METH_NOARGS, "calls / misses so far"},
    {NULL},
};

static PyTypeObject FastPlanType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "pytensor_amd._fastplan.FastPlan",
    .tp_basicsize = sizeof(FastPlan),
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_new = PyType_GenericNew,
    .tp_init = (initproc)FastPlan_init,
    .tp_dealloc = (destructor)FastPlan_dealloc,
    .tp_call = (ternaryfunc)FastPlan_call,
    .tp_methods = FastPlan_methods,
    .tp_doc = "the replay path of a frozen plan as one native call (see csrc/fastplan.c)",
};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastplan", "native call path of pytensor_amd.plan.FrozenPlan", -1, NULL};

PyMODINIT_FUNC PyInit__fastplan(void) {
  import_array();
  if (PyType_Ready(&FastPlanType) < 0) return NULL;
  PyObject* m = PyModule_Create(&moddef);
  if (!m) return NULL;
  Py_INCREF(&FastPlanType);
  if (PyModule_AddObject(m, "FastPlan", (PyObject*)&FastPlanType) < 0) { Py_DECREF(&FastPlanType); Py_DECREF(m); return NULL; }
  return m;
}
