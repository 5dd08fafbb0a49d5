/*
 * N-API addon: the JavaScript face of include/ca3d.h, 1:1, typed arrays in and out. This is the binding a
 * maintainer of main_pathtraced.js adds in place of the navigator.gpu calls. No logic lives
 * here: every function forwards to libca3d.so and throws a JS Error carrying ca3d_last_error() on failure.
 *
 * Built with plain gcc against /usr/include/node (N-API 8 headers; Node >= 12).
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ca3d.h"

#define NAPI_OK_OR_NULL(call)                                  \
	do {                                                       \
		if ((call) != napi_ok) {                               \
			napi_throw_error(env, NULL, "N-API call failed: " #call); \
			return NULL;                                       \
		}                                                      \
	} while (0)

static napi_value throw_ca3d(napi_env env, int rc)
{
	char msg[600];
	snprintf(msg, sizeof msg, "ca3d error %d: %s", rc, ca3d_last_error());
	napi_throw_error(env, NULL, msg);
	return NULL;
}

static napi_value undefined(napi_env env)
{
	napi_value u;
	napi_get_undefined(env, &u);
	return u;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv)
{
	size_t argc = want;
	if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want)
	{
		napi_throw_type_error(env, NULL, "wrong number of arguments");
		return 0;
	}
	return 1;
}

/* What a handle external points at: the engine and the number of asynchronous jobs that still use it (main thread only). */
typedef struct
{
	ca3d_t *h; /* first member: a Slot * reads as a ca3d_t ** */
	int pending;
} Slot;

static ca3d_t *get_handle(napi_env env, napi_value v)
{
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p)
	{
		napi_throw_type_error(env, NULL, "expected an engine handle");
		return NULL;
	}
	return *(ca3d_t **)p;
}

static int get_u32(napi_env env, napi_value v, uint32_t *out)
{
	if (napi_get_value_uint32(env, v, out) != napi_ok)
	{
		napi_throw_type_error(env, NULL, "expected an unsigned integer");
		return 0;
	}
	return 1;
}

/* typed array of the given element type, or null/undefined when `nullable` */
static int get_typed(napi_env env, napi_value v, napi_typedarray_type want, int nullable, void **data, size_t *len)
{
	napi_valuetype t;
	napi_typeof(env, v, &t);
	if (nullable && (t == napi_null || t == napi_undefined))
	{
		*data = NULL;
		*len = 0;
		return 1;
	}
	bool is = false;
	napi_is_typedarray(env, v, &is);
	napi_typedarray_type type;
	napi_value ab;
	size_t off;
	if (!is || napi_get_typedarray_info(env, v, &type, len, data, &ab, &off) != napi_ok || type != want)
	{
		napi_throw_type_error(env, NULL, "expected a typed array of the documented element type");
		return 0;
	}
	return 1;
}

static void finalize_handle(napi_env env, void *data, void *hint)
{
	(void)env;
	(void)hint;
	Slot *slot = (Slot *)data; /* no job is pending: every job holds a reference to the external */
	if (slot->h) ca3d_destroy(slot->h);
	slot->h = NULL;
	free(slot);
}

static napi_value js_create(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	int32_t device = 0;
	napi_get_value_int32(env, argv[0], &device);
	ca3d_t *h = NULL;
	int rc = ca3d_create(device, &h);
	if (rc) return throw_ca3d(env, rc);
	Slot *slot = (Slot *)calloc(1, sizeof *slot);
	if (!slot) { ca3d_destroy(h); napi_throw_error(env, NULL, "out of memory"); return NULL; }
	slot->h = h;
	napi_value ext;
	NAPI_OK_OR_NULL(napi_create_external(env, slot, finalize_handle, NULL, &ext));
	return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	void *p = NULL;
	if (napi_get_value_external(env, argv[0], &p) == napi_ok && p)
	{
		Slot *slot = (Slot *)p;
		if (slot->pending)
		{
			/* a libuv worker is inside ca3d_read_state / ca3d_render / ca3d_synchronize on this engine */
			napi_throw_error(env, NULL, "the engine has asynchronous work pending: await it before close()");
			return NULL;
		}
		if (slot->h) ca3d_destroy(slot->h);
		slot->h = NULL;
	}
	return undefined(env);
}

static napi_value js_device_count(napi_env env, napi_callback_info info)
{
	(void)info;
	int n = 0;
	int rc = ca3d_device_count(&n);
	if (rc) return throw_ca3d(env, rc);
	napi_value v;
	napi_create_int32(env, n, &v);
	return v;
}

static napi_value js_configure(napi_env env, napi_callback_info info)
{
	napi_value argv[3];
	if (!get_args(env, info, 3, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	uint32_t g, layout;
	if (!h || !get_u32(env, argv[1], &g) || !get_u32(env, argv[2], &layout)) return NULL;
	int rc = ca3d_configure(h, g, g, g, (int)layout);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_configure_slab(napi_env env, napi_callback_info info)
{
	napi_value argv[6];
	if (!get_args(env, info, 6, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	uint32_t g, layout, z0, nz, ghost;
	if (!h || !get_u32(env, argv[1], &g) || !get_u32(env, argv[2], &layout) || !get_u32(env, argv[3], &z0) ||
	    !get_u32(env, argv[4], &nz) || !get_u32(env, argv[5], &ghost))
		return NULL;
	int rc = ca3d_configure_slab(h, g, (int)layout, z0, nz, ghost);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_set_rules(napi_env env, napi_callback_info info)
{
	napi_value argv[6];
	if (!get_args(env, info, 6, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	void *m, *e, *c, *s, *b;
	size_t nm, ne, nc, ns, nb;
	if (!get_typed(env, argv[1], napi_int32_array, 0, &m, &nm) || !get_typed(env, argv[2], napi_int32_array, 0, &e, &ne) ||
	    !get_typed(env, argv[3], napi_int32_array, 0, &c, &nc) || !get_typed(env, argv[4], napi_uint32_array, 0, &s, &ns) ||
	    !get_typed(env, argv[5], napi_uint32_array, 0, &b, &nb))
		return NULL;
	if (ns != CA3D_LUT_LEN || nb != CA3D_LUT_LEN)
	{
		napi_throw_range_error(env, NULL, "survive/born must be Uint32Array(81)");
		return NULL;
	}
	int rc = ca3d_set_rules(h, (const int32_t *)m, (uint32_t)nm, (const int32_t *)e, (uint32_t)ne, (const int32_t *)c,
	                        (uint32_t)nc, (const uint32_t *)s, (const uint32_t *)b);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_upload_state(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	void *w;
	size_t n;
	if (!h || !get_typed(env, argv[1], napi_uint32_array, 0, &w, &n)) return NULL;
	int rc = ca3d_upload_state(h, (const uint32_t *)w, n);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_read_state(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	void *w;
	size_t n;
	if (!h || !get_typed(env, argv[1], napi_uint32_array, 0, &w, &n)) return NULL;
	int rc = ca3d_read_state(h, (uint32_t *)w, n);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_step(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	uint32_t n;
	if (!h || !get_u32(env, argv[1], &n)) return NULL;
	int rc = ca3d_step(h, n);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_flush(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	int rc = ca3d_flush(h);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_slab_step(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	uint32_t n;
	if (!h || !get_u32(env, argv[1], &n)) return NULL;
	int rc = ca3d_slab_step(h, n);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_slab_step_phase(napi_env env, napi_callback_info info)
{
	napi_value argv[3];
	if (!get_args(env, info, 3, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	uint32_t n, phase;
	if (!h || !get_u32(env, argv[1], &n) || !get_u32(env, argv[2], &phase)) return NULL;
	int rc = ca3d_slab_step_phase(h, n, (int)phase);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_synchronize(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	int rc = ca3d_synchronize(h);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static void set_num(napi_env env, napi_value obj, const char *k, double v)
{
	napi_value n;
	napi_create_double(env, v, &n);
	napi_set_named_property(env, obj, k, n);
}

static napi_value js_info(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	ca3d_info i;
	int rc = ca3d_get_info(h, &i);
	if (rc) return throw_ca3d(env, rc);
	napi_value o, s;
	napi_create_object(env, &o);
	set_num(env, o, "gridSize", i.grid_size);
	set_num(env, o, "layout", i.layout);
	set_num(env, o, "z0", i.z0);
	set_num(env, o, "nz", i.nz);
	set_num(env, o, "ghost", i.ghost);
	set_num(env, o, "step", (double)i.step);
	set_num(env, o, "stateWords", (double)i.state_words);
	set_num(env, o, "currentBuffer", i.current_buffer);
	set_num(env, o, "device", i.device);
	set_num(env, o, "launchesTotal", (double)i.launches_total);
	napi_create_string_utf8(env, i.kernel_name, NAPI_AUTO_LENGTH, &s);
	napi_set_named_property(env, o, "kernelName", s);
	char variant[384];
	if (ca3d_get_kernel_variant(h, variant, sizeof variant, NULL) == CA3D_OK)
	{
		napi_create_string_utf8(env, variant, NAPI_AUTO_LENGTH, &s);
		napi_set_named_property(env, o, "kernelVariant", s);
	}
	return o;
}

static napi_value js_stats(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	ca3d_stats st;
	int rc = ca3d_get_stats(h, &st);
	if (rc) return throw_ca3d(env, rc);
	napi_value o;
	napi_create_object(env, &o);
	set_num(env, o, "steps", (double)st.steps);
	set_num(env, o, "kernelLaunches", (double)st.kernel_launches);
	set_num(env, o, "gpuMs", st.gpu_ms);
	set_num(env, o, "cellSteps", st.cell_steps);
	set_num(env, o, "algorithmicBytes", st.algorithmic_bytes);
	return o;
}

static napi_value js_render(napi_env env, napi_callback_info info)
{
	napi_value argv[8];
	if (!get_args(env, info, 8, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	void *u, *pres, *light, *depth;
	size_t nu, npres, nlight, ndepth;
	uint32_t w, hh, spp;
	if (!get_typed(env, argv[1], napi_float32_array, 0, &u, &nu) || !get_u32(env, argv[2], &w) || !get_u32(env, argv[3], &hh) ||
	    !get_u32(env, argv[4], &spp) || !get_typed(env, argv[5], napi_uint8_array, 1, &pres, &npres) ||
	    !get_typed(env, argv[6], napi_uint16_array, 1, &light, &nlight) || !get_typed(env, argv[7], napi_uint16_array, 1, &depth, &ndepth))
		return NULL;
	const size_t px = (size_t)w * hh;
	if (nu != 128 || (pres && npres != px * 4) || (light && nlight != px * 4) || (depth && ndepth != px * 2))
	{
		napi_throw_range_error(env, NULL, "uniforms must be Float32Array(128); targets must match width*height");
		return NULL;
	}
	int rc = ca3d_render(h, (const float *)u, w, hh, spp, (uint8_t *)pres, (uint16_t *)light, (uint16_t *)depth);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_render_stats(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	ca3d_render_stats st;
	int rc = ca3d_get_render_stats(h, &st);
	if (rc) return throw_ca3d(env, rc);
	napi_value o;
	napi_create_object(env, &o);
	set_num(env, o, "gpuMs", st.gpu_ms);
	set_num(env, o, "primaryRays", (double)st.primary_rays);
	set_num(env, o, "shadowRays", (double)st.shadow_rays);
	set_num(env, o, "primaryCellVisits", (double)st.primary_cell_visits);
	set_num(env, o, "shadowCellVisits", (double)st.shadow_cell_visits);
	return o;
}

static napi_value js_render_pipeline(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	int32_t n = 0;
	int rc = ca3d_get_render_pipeline(h, &n);
	if (rc) return throw_ca3d(env, rc);
	napi_value v;
	napi_create_int32(env, n, &v);
	return v;
}

static napi_value js_set_option(napi_env env, napi_callback_info info)
{
	napi_value argv[3];
	if (!get_args(env, info, 3, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	char name[64];
	size_t len = 0;
	int64_t v = 0;
	napi_get_value_string_utf8(env, argv[1], name, sizeof name, &len);
	napi_get_value_int64(env, argv[2], &v);
	int rc = ca3d_set_option(h, name, v);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

/* RCCL transport inside the engine (include/ca3d.h "Halo transport"): a Node.js host runs one process per GPU, creates the
 * id on rank 0, hands the 128 bytes to the other ranks over its own IPC, and calls slabRun — no Python involved. */
static napi_value js_comm_unique_id(napi_env env, napi_callback_info info)
{
	(void)info;
	void *data = NULL;
	napi_value ab, out;
	NAPI_OK_OR_NULL(napi_create_arraybuffer(env, CA3D_COMM_ID_BYTES, &data, &ab));
	int rc = ca3d_comm_unique_id(data);
	if (rc) return throw_ca3d(env, rc);
	NAPI_OK_OR_NULL(napi_create_typedarray(env, napi_uint8_array, CA3D_COMM_ID_BYTES, ab, 0, &out));
	return out;
}

static napi_value js_slab_comm_init(napi_env env, napi_callback_info info)
{
	napi_value argv[4];
	if (!get_args(env, info, 4, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	void *id;
	size_t n;
	uint32_t rank, world;
	if (!h || !get_typed(env, argv[1], napi_uint8_array, 0, &id, &n) || !get_u32(env, argv[2], &rank) || !get_u32(env, argv[3], &world)) return NULL;
	if (n != CA3D_COMM_ID_BYTES) { napi_throw_range_error(env, NULL, "the communicator id is a Uint8Array(128)"); return NULL; }
	int rc = ca3d_slab_comm_init(h, id, (int)rank, (int)world);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_slab_run(napi_env env, napi_callback_info info)
{
	napi_value argv[3];
	if (!get_args(env, info, 3, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	uint32_t n, overlap;
	if (!h || !get_u32(env, argv[1], &n) || !get_u32(env, argv[2], &overlap)) return NULL;
	int rc = ca3d_slab_run(h, n, (int)overlap);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_slab_exchange(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	int rc = ca3d_slab_exchange(h);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_slab_gather(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]), *full = get_handle(env, argv[1]);
	if (!h || !full) return NULL;
	int rc = ca3d_slab_gather(h, full);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

/*
 * Asynchronous forms of the calls that wait for the GPU ("optional napi_async_work wrappers"): the wait
 * runs on a libuv worker thread and the call returns a Promise, so a UI thread never blocks in a read-back. The engine
 * is not thread-safe: the JS wrapper (ca3d.js) queues everything else behind a pending job of the same engine.
 */
typedef struct
{
	napi_async_work work;
	napi_deferred deferred;
	ca3d_t *h;
	int kind; /* 0 readState, 1 render, 2 synchronize */
	int rc;
	char err[512];
	uint32_t *words;
	size_t n_words;
	float uniforms[128];
	uint32_t w, hh, spp;
	uint8_t *pres;
	uint16_t *light, *depth;
	napi_ref refs[4]; /* the handle external and the typed arrays stay alive (and in place) until the job completes */
	int nrefs;
	Slot *slot;       /* its `pending` count keeps destroy() away while the worker runs */
} AsyncJob;

static void job_execute(napi_env env, void *data)
{
	(void)env;
	AsyncJob *j = (AsyncJob *)data;
	if (j->kind == 0) j->rc = ca3d_read_state(j->h, j->words, j->n_words);
	else if (j->kind == 1) j->rc = ca3d_render(j->h, j->uniforms, j->w, j->hh, j->spp, j->pres, j->light, j->depth);
	else j->rc = ca3d_synchronize(j->h);
	if (j->rc) snprintf(j->err, sizeof j->err, "ca3d error %d: %s", j->rc, ca3d_last_error()); /* ca3d_last_error is per thread */
}

static void job_complete(napi_env env, napi_status status, void *data)
{
	AsyncJob *j = (AsyncJob *)data;
	napi_value v;
	if (status == napi_ok && j->rc == 0)
	{
		napi_get_undefined(env, &v);
		napi_resolve_deferred(env, j->deferred, v);
	}
	else
	{
		napi_value msg;
		napi_create_string_utf8(env, status == napi_ok ? j->err : "the asynchronous job was cancelled", NAPI_AUTO_LENGTH, &msg);
		napi_create_error(env, NULL, msg, &v);
		napi_reject_deferred(env, j->deferred, v);
	}
	if (j->slot) j->slot->pending--;
	for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
	napi_delete_async_work(env, j->work);
	free(j);
}

static napi_value job_start(napi_env env, AsyncJob *j, const char *name)
{
	napi_value promise, resource;
	if (napi_create_promise(env, &j->deferred, &promise) != napi_ok || napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &resource) != napi_ok ||
	    napi_create_async_work(env, NULL, resource, job_execute, job_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok)
	{
		for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
		free(j);
		napi_throw_error(env, NULL, "could not queue the asynchronous job");
		return NULL;
	}
	if (j->slot) j->slot->pending++;
	return promise;
}

static void job_keep(napi_env env, AsyncJob *j, napi_value v)
{
	napi_valuetype t;
	napi_typeof(env, v, &t);
	if (t == napi_null || t == napi_undefined) return;
	if (j->nrefs < (int)(sizeof j->refs / sizeof j->refs[0]) && napi_create_reference(env, v, 1, &j->refs[j->nrefs]) == napi_ok) j->nrefs++;
}

/* The job keeps the handle external alive (its finalizer destroys the engine) and counts itself on the slot. */
static void job_hold_engine(napi_env env, AsyncJob *j, napi_value handle)
{
	void *p = NULL;
	if (napi_get_value_external(env, handle, &p) == napi_ok) j->slot = (Slot *)p;
	job_keep(env, j, handle);
}

static napi_value js_read_state_async(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	void *w;
	size_t n;
	if (!h || !get_typed(env, argv[1], napi_uint32_array, 0, &w, &n)) return NULL;
	AsyncJob *j = (AsyncJob *)calloc(1, sizeof *j);
	if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
	j->h = h; j->kind = 0; j->words = (uint32_t *)w; j->n_words = n;
	job_hold_engine(env, j, argv[0]);
	job_keep(env, j, argv[1]);
	return job_start(env, j, "ca3d.readState");
}

static napi_value js_render_async(napi_env env, napi_callback_info info)
{
	napi_value argv[8];
	if (!get_args(env, info, 8, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	void *u, *pres, *light, *depth;
	size_t nu, npres, nlight, ndepth;
	uint32_t w, hh, spp;
	if (!get_typed(env, argv[1], napi_float32_array, 0, &u, &nu) || !get_u32(env, argv[2], &w) || !get_u32(env, argv[3], &hh) ||
	    !get_u32(env, argv[4], &spp) || !get_typed(env, argv[5], napi_uint8_array, 1, &pres, &npres) ||
	    !get_typed(env, argv[6], napi_uint16_array, 1, &light, &nlight) || !get_typed(env, argv[7], napi_uint16_array, 1, &depth, &ndepth))
		return NULL;
	const size_t px = (size_t)w * hh;
	if (nu != 128 || (pres && npres != px * 4) || (light && nlight != px * 4) || (depth && ndepth != px * 2))
	{
		napi_throw_range_error(env, NULL, "uniforms must be Float32Array(128); targets must match width*height");
		return NULL;
	}
	AsyncJob *j = (AsyncJob *)calloc(1, sizeof *j);
	if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
	j->h = h; j->kind = 1; j->w = w; j->hh = hh; j->spp = spp;
	memcpy(j->uniforms, u, sizeof j->uniforms); /* the block is consumed now, like the reference's writeBuffer */
	j->pres = (uint8_t *)pres; j->light = (uint16_t *)light; j->depth = (uint16_t *)depth;
	job_hold_engine(env, j, argv[0]);
	job_keep(env, j, argv[5]); job_keep(env, j, argv[6]); job_keep(env, j, argv[7]);
	return job_start(env, j, "ca3d.render");
}

static napi_value js_synchronize_async(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	AsyncJob *j = (AsyncJob *)calloc(1, sizeof *j);
	if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
	j->h = h; j->kind = 2;
	job_hold_engine(env, j, argv[0]);
	return job_start(env, j, "ca3d.synchronize");
}

/* ---- ca3d_group_*: one JS thread drives the Z-slab split over several GPUs (include/ca3d.h) ---------------------------- */
typedef struct
{
	ca3d_group_t *g;
} GroupSlot;

static void finalize_group(napi_env env, void *data, void *hint)
{
	(void)env;
	(void)hint;
	GroupSlot *slot = (GroupSlot *)data;
	if (slot->g) ca3d_group_destroy(slot->g);
	free(slot);
}

static ca3d_group_t *get_group(napi_env env, napi_value v)
{
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((GroupSlot *)p)->g)
	{
		napi_throw_type_error(env, NULL, "expected an engine-group handle");
		return NULL;
	}
	return ((GroupSlot *)p)->g;
}

static napi_value js_group_create(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	void *d;
	size_t n;
	if (!get_typed(env, argv[0], napi_int32_array, 0, &d, &n)) return NULL;
	ca3d_group_t *g = NULL;
	int rc = ca3d_group_create((const int *)d, (int)n, &g);
	if (rc) return throw_ca3d(env, rc);
	GroupSlot *slot = (GroupSlot *)calloc(1, sizeof *slot);
	if (!slot) { ca3d_group_destroy(g); napi_throw_error(env, NULL, "out of memory"); return NULL; }
	slot->g = g;
	napi_value ext;
	NAPI_OK_OR_NULL(napi_create_external(env, slot, finalize_group, NULL, &ext));
	return ext;
}

static napi_value js_group_destroy(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	void *p = NULL;
	if (napi_get_value_external(env, argv[0], &p) == napi_ok && p)
	{
		GroupSlot *slot = (GroupSlot *)p;
		if (slot->g) ca3d_group_destroy(slot->g);
		slot->g = NULL;
	}
	return undefined(env);
}

static napi_value js_group_configure(napi_env env, napi_callback_info info)
{
	napi_value argv[4];
	if (!get_args(env, info, 4, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	uint32_t grid, layout, ghost;
	if (!g || !get_u32(env, argv[1], &grid) || !get_u32(env, argv[2], &layout) || !get_u32(env, argv[3], &ghost)) return NULL;
	int rc = ca3d_group_configure(g, grid, (int)layout, ghost);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_group_set_rules(napi_env env, napi_callback_info info)
{
	napi_value argv[6];
	if (!get_args(env, info, 6, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	if (!g) return NULL;
	void *m, *e, *c, *s, *b;
	size_t nm, ne, nc, ns, nb;
	if (!get_typed(env, argv[1], napi_int32_array, 0, &m, &nm) || !get_typed(env, argv[2], napi_int32_array, 0, &e, &ne) ||
	    !get_typed(env, argv[3], napi_int32_array, 0, &c, &nc) || !get_typed(env, argv[4], napi_uint32_array, 0, &s, &ns) ||
	    !get_typed(env, argv[5], napi_uint32_array, 0, &b, &nb))
		return NULL;
	if (ns != CA3D_LUT_LEN || nb != CA3D_LUT_LEN)
	{
		napi_throw_range_error(env, NULL, "survive/born must be Uint32Array(81)");
		return NULL;
	}
	int rc = ca3d_group_set_rules(g, (const int32_t *)m, (uint32_t)nm, (const int32_t *)e, (uint32_t)ne, (const int32_t *)c, (uint32_t)nc,
	                              (const uint32_t *)s, (const uint32_t *)b);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_group_upload_state(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	void *w;
	size_t n;
	if (!g || !get_typed(env, argv[1], napi_uint32_array, 0, &w, &n)) return NULL;
	int rc = ca3d_group_upload_state(g, (const uint32_t *)w, n);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_group_read_state(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	void *w;
	size_t n;
	if (!g || !get_typed(env, argv[1], napi_uint32_array, 0, &w, &n)) return NULL;
	int rc = ca3d_group_read_state(g, (uint32_t *)w, n);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_group_step(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	uint32_t n;
	if (!g || !get_u32(env, argv[1], &n)) return NULL;
	int rc = ca3d_group_step(g, n);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_group_synchronize(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	if (!g) return NULL;
	int rc = ca3d_group_synchronize(g);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_group_set_option(napi_env env, napi_callback_info info)
{
	napi_value argv[3];
	if (!get_args(env, info, 3, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	if (!g) return NULL;
	char name[64];
	size_t len = 0;
	int64_t value = 0;
	if (napi_get_value_string_utf8(env, argv[1], name, sizeof name, &len) != napi_ok || napi_get_value_int64(env, argv[2], &value) != napi_ok)
	{
		napi_throw_type_error(env, NULL, "expected (group, name, integer)");
		return NULL;
	}
	int rc = ca3d_group_set_option(g, name, value);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

/* info of rank's slab engine (kernel name, planes, step ...) */
static napi_value js_group_info(napi_env env, napi_callback_info info)
{
	napi_value argv[2];
	if (!get_args(env, info, 2, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	uint32_t rank;
	if (!g || !get_u32(env, argv[1], &rank)) return NULL;
	ca3d_t *e = NULL;
	int rc = ca3d_group_engine(g, (int)rank, &e);
	ca3d_info i;
	if (rc == 0) rc = ca3d_get_info(e, &i);
	if (rc) return throw_ca3d(env, rc);
	int n = 0;
	ca3d_group_size(g, &n);
	napi_value o, s;
	napi_create_object(env, &o);
	set_num(env, o, "ranks", n);
	set_num(env, o, "gridSize", i.grid_size);
	set_num(env, o, "z0", i.z0);
	set_num(env, o, "nz", i.nz);
	set_num(env, o, "ghost", i.ghost);
	set_num(env, o, "step", (double)i.step);
	set_num(env, o, "device", i.device);
	set_num(env, o, "launchesTotal", (double)i.launches_total);
	napi_create_string_utf8(env, i.kernel_name, NAPI_AUTO_LENGTH, &s);
	napi_set_named_property(env, o, "kernelName", s);
	/* what the slab really runs on (ca3d_slab_comm_info): the PCI bus id tells two GPUs from one GPU listed twice */
	ca3d_comm_info ci;
	if (ca3d_slab_comm_info(e, &ci) == CA3D_OK)
	{
		napi_create_string_utf8(env, ci.pci_bus_id, NAPI_AUTO_LENGTH, &s);
		napi_set_named_property(env, o, "pciBusId", s);
		set_num(env, o, "commRanks", ci.comm_ranks);
	}
	return o;
}

static napi_value js_group_render(napi_env env, napi_callback_info info)
{
	napi_value argv[8];
	if (!get_args(env, info, 8, argv)) return NULL;
	ca3d_group_t *g = get_group(env, argv[0]);
	if (!g) return NULL;
	void *u, *pres, *light, *depth;
	size_t nu, npres, nlight, ndepth;
	uint32_t w, hh, spp;
	if (!get_typed(env, argv[1], napi_float32_array, 0, &u, &nu) || !get_u32(env, argv[2], &w) || !get_u32(env, argv[3], &hh) ||
	    !get_u32(env, argv[4], &spp) || !get_typed(env, argv[5], napi_uint8_array, 1, &pres, &npres) ||
	    !get_typed(env, argv[6], napi_uint16_array, 1, &light, &nlight) || !get_typed(env, argv[7], napi_uint16_array, 1, &depth, &ndepth))
		return NULL;
	const size_t px = (size_t)w * hh;
	if (nu != 128 || (pres && npres != px * 4) || (light && nlight != px * 4) || (depth && ndepth != px * 2))
	{
		napi_throw_range_error(env, NULL, "uniforms must be Float32Array(128); targets must match width*height");
		return NULL;
	}
	int rc = ca3d_group_render(g, (const float *)u, w, hh, spp, (uint8_t *)pres, (uint16_t *)light, (uint16_t *)depth);
	return rc ? throw_ca3d(env, rc) : undefined(env);
}

static napi_value js_recovered_launches(napi_env env, napi_callback_info info)
{
	napi_value argv[1];
	if (!get_args(env, info, 1, argv)) return NULL;
	ca3d_t *h = get_handle(env, argv[0]);
	if (!h) return NULL;
	uint32_t n = 0;
	int rc = ca3d_recovered_launches(h, &n);
	if (rc) return throw_ca3d(env, rc);
	napi_value v;
	napi_create_uint32(env, n, &v);
	return v;
}

static napi_value js_abi_version(napi_env env, napi_callback_info info)
{
	(void)info;
	napi_value v;
	napi_create_int32(env, ca3d_abi_version(), &v);
	return v;
}

/* ca3d_get_jit_stats: what the run-time compiler did in this process (compiled / read from the on-disk cache / already loaded) */
static napi_value js_jit_stats(napi_env env, napi_callback_info info)
{
	(void)info;
	ca3d_jit_stats st;
	int rc = ca3d_get_jit_stats(&st);
	if (rc != CA3D_OK) return throw_ca3d(env, rc);
	napi_value o, s;
	napi_create_object(env, &o);
	set_num(env, o, "programsCompiled", (double)st.programs_compiled);
	set_num(env, o, "programsFromDisk", (double)st.programs_from_disk);
	set_num(env, o, "programsFromMemory", (double)st.programs_from_memory);
	set_num(env, o, "compileMs", st.compile_ms);
	set_num(env, o, "diskReadMs", st.disk_read_ms);
	set_num(env, o, "loadMs", st.load_ms);
	napi_create_string_utf8(env, st.cache_dir, NAPI_AUTO_LENGTH, &s);
	napi_set_named_property(env, o, "cacheDir", s);
	return o;
}

static napi_value init(napi_env env, napi_value exports)
{
	static const struct { const char *name; napi_callback fn; } fns[] = {
	    {"abiVersion", js_abi_version}, {"jitStats", js_jit_stats}, {"deviceCount", js_device_count}, {"create", js_create}, {"destroy", js_destroy},
	    {"configure", js_configure}, {"configureSlab", js_configure_slab}, {"setRules", js_set_rules},
	    {"uploadState", js_upload_state}, {"readState", js_read_state}, {"step", js_step}, {"flush", js_flush}, {"slabStep", js_slab_step}, {"slabStepPhase", js_slab_step_phase},
	    {"synchronize", js_synchronize}, {"info", js_info}, {"stats", js_stats}, {"render", js_render},
	    {"renderStats", js_render_stats}, {"renderPipeline", js_render_pipeline}, {"setOption", js_set_option},
	    {"commUniqueId", js_comm_unique_id}, {"slabCommInit", js_slab_comm_init}, {"slabRun", js_slab_run}, {"slabExchange", js_slab_exchange},
	    {"slabGather", js_slab_gather},
	    {"recoveredLaunches", js_recovered_launches},
	    {"groupCreate", js_group_create}, {"groupDestroy", js_group_destroy}, {"groupConfigure", js_group_configure}, {"groupSetRules", js_group_set_rules},
	    {"groupUploadState", js_group_upload_state}, {"groupReadState", js_group_read_state}, {"groupStep", js_group_step},
	    {"groupSynchronize", js_group_synchronize}, {"groupSetOption", js_group_set_option}, {"groupInfo", js_group_info}, {"groupRender", js_group_render},
	    {"readStateAsync", js_read_state_async}, {"renderAsync", js_render_async}, {"synchronizeAsync", js_synchronize_async}};
	for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++)
	{
		napi_value f;
		if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
		napi_set_named_property(env, exports, fns[i].name, f);
	}
	return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
