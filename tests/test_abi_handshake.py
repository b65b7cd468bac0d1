"""ABI handshake (include/pbrt_mi355.h): every entry point that reads or writes a struct takes the caller's sizeof(); a caller
built against another revision of the header gets PM_ERR_INVALID and a message instead of a crash.  (Round 2's GPU log holds a
SIGSEGV of examples/main_rs that had been compiled before PmLight / PmMaterial grew and ran against the rebuilt library.)
Host-only contexts (device -1) reach every check below without a GPU."""
import ctypes as C
import re
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctx(pm):
    lib = pm._capi.lib()
    h = C.c_void_p()
    assert lib.pm_create(-1, C.byref(h)) == 0
    return lib, h


def test_abi_version_matches_header(pm):
    lib = pm._capi.lib()
    text = open(os.path.join(ROOT, "include", "pbrt_mi355.h")).read()
    ver = int(re.search(r"#define PM_ABI_VERSION (\d+)u", text).group(1))
    assert lib.pm_abi_version() == ver == pm._capi.PM_ABI_VERSION
    rs = open(os.path.join(ROOT, "rust", "pbrt-mi355-sys", "src", "lib.rs")).read()
    assert int(re.search(r"pub const PM_ABI_VERSION: u32 = (\d+);", rs).group(1)) == ver


def test_wrong_struct_size_is_rejected_not_read(pm):
    cap = pm._capi
    lib, h = _ctx(pm)
    try:
        # a stale caller's PmLight (round 2: 80 bytes before prim_id / two_sided were added): rejected before any field is read
        spectrum = (C.c_float * 471)(*([1.0] * 471))
        lights = (cap.PmLight * 1)()
        lights[0].type = cap.PM_LIGHT_POINT
        lights[0].spectrum = C.cast(spectrum, C.POINTER(C.c_float))
        assert lib.pm_set_lights(h, lights, 1, 80) == cap.PM_ERR_INVALID
        msg = lib.pm_last_error(h).decode()
        assert "struct size mismatch" in msg and "PmLight" in msg and "80" in msg and str(C.sizeof(cap.PmLight)) in msg
        assert lib.pm_set_lights(h, lights, 1, C.sizeof(cap.PmLight)) == cap.PM_OK
        mats = (cap.PmMaterial * 1)()
        assert lib.pm_set_materials(h, mats, 1, C.sizeof(cap.PmMaterial) - 8) == cap.PM_ERR_INVALID
        assert lib.pm_set_materials(h, mats, 1, C.sizeof(cap.PmMaterial)) == cap.PM_OK
        cam = cap.PmCamera()
        assert lib.pm_set_camera(h, C.byref(cam), 0) == cap.PM_ERR_INVALID
        film = cap.PmFilm()
        assert lib.pm_set_film(h, C.byref(film), C.sizeof(film) + 8) == cap.PM_ERR_INVALID
        assert "PmFilm" in lib.pm_last_error(h).decode()
        sph = cap.PmSphere()
        first = C.c_uint32()
        assert lib.pm_add_spheres(h, C.byref(sph), 1, C.byref(first), 148) == cap.PM_ERR_INVALID      # (round 1's excerpt omitted _pad)
        p = cap.PmRenderParams()
        assert lib.pm_render(h, C.byref(p), 44) == cap.PM_ERR_INVALID
        # output structs: a smaller caller struct would be overrun by the library's write
        pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
        idx = np.array([[0, 1, 2]], np.uint32)
        assert lib.pm_add_triangle_mesh(h, pos.ctypes.data, 3, idx.ctypes.data, 1, C.byref(first)) == cap.PM_OK
        info = cap.PmAccelInfo()
        assert lib.pm_build_accel(h, cap.PM_SPLIT_SAH, 4, C.byref(info), C.sizeof(info) - 8) == cap.PM_ERR_INVALID
        assert lib.pm_build_accel(h, cap.PM_SPLIT_SAH, 4, C.byref(info), C.sizeof(info)) == cap.PM_OK and info.n_prims == 1
        assert lib.pm_build_accel(h, cap.PM_SPLIT_SAH, 4, None, 0) == cap.PM_OK                         # no struct, no check
    finally:
        lib.pm_destroy(h)


def test_rejected_medium_and_materials_leave_the_previous_state(pm):
    """pm_set_medium committed state before it finished validating."""
    cap = pm._capi
    lib, h = _ctx(pm)
    try:
        good_a = (C.c_float * 471)(*([0.05] * 471))
        good_s = (C.c_float * 471)(*([1.0] * 471))
        bad_s = (C.c_float * 471)(*([1.0] * 470 + [-1.0]))
        m = cap.PmMedium()
        m.sigma_a = C.cast(good_a, C.POINTER(C.c_float)); m.sigma_s = C.cast(good_s, C.POINTER(C.c_float)); m.g = 0.6
        m.bounds[:] = [-1, -1, -1, 1, 1, 1]
        assert lib.pm_set_medium(h, C.byref(m), C.sizeof(m)) == cap.PM_OK
        state = (C.c_float * 4)()
        assert lib.pm_debug_medium_state(h, state) == cap.PM_OK and state[0] == 1.0 and abs(state[1] - 0.6) < 1e-7 and state[3] == 1.0
        m2 = cap.PmMedium.from_buffer_copy(m)
        m2.sigma_s = C.cast(bad_s, C.POINTER(C.c_float)); m2.g = 0.1
        assert lib.pm_set_medium(h, C.byref(m2), C.sizeof(m2)) == cap.PM_ERR_INVALID
        assert lib.pm_debug_medium_state(h, state) == cap.PM_OK
        assert state[0] == 1.0 and abs(state[1] - 0.6) < 1e-7 and state[3] == 1.0, "the rejected medium replaced the good one"
        nan_a = (C.c_float * 471)(*([float("nan")] * 471))
        m2.sigma_s = m.sigma_s; m2.sigma_a = C.cast(nan_a, C.POINTER(C.c_float))
        assert lib.pm_set_medium(h, C.byref(m2), C.sizeof(m2)) == cap.PM_ERR_INVALID
        assert lib.pm_debug_medium_state(h, state) == cap.PM_OK and abs(state[1] - 0.6) < 1e-7
        assert lib.pm_set_medium(h, None, 0) == cap.PM_OK
        assert lib.pm_debug_medium_state(h, state) == cap.PM_OK and state[0] == 0.0
    finally:
        lib.pm_destroy(h)
