""".pbrt ingestion (pbrt-rs_amd/pbrt_scene.py): tests/golden/main_rs.pbrt is the reference binary's hard-coded scene
(src/main.rs:43-234) written as a scene file; parsing it must give the flattened scene the hard-coded builder gives."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _same_scene(pm, sd, ref):
    assert np.array_equal(np.array(sd.camera.camera_from_raster[:]), np.array(ref.camera.camera_from_raster[:]))
    assert np.array_equal(np.array(sd.camera.render_from_camera[:]), np.array(ref.camera.render_from_camera[:]))
    assert len(sd.lights) == len(ref.lights)
    for a, b in zip(sd.lights, ref.lights):
        assert a["type"] == b["type"] and a["scale"] == b["scale"] and np.array_equal(a["table"], b["table"])
        if a["type"] != pm.host.PM_LIGHT_UNIFORM_INFINITE:
            assert np.array_equal(a["m"], b["m"])
    assert [(m["type"], m["spectrum"], m["remap"]) for m in sd.materials] == [(m["type"], m["spectrum"], m["remap"]) for m in ref.materials]
    assert list(sd.prim_materials) == list(ref.prim_materials)
    for a, b in zip(sd.prims, ref.prims):
        assert type(a) is type(b)
        if isinstance(a, pm.Sphere):
            assert (a.radius, a.z_min, a.z_max, a.phi_max) == (b.radius, b.z_min, b.z_max, b.phi_max)
            assert np.array_equal(a.render_from_object, b.render_from_object)
        else:
            assert np.array_equal(a.positions, b.positions) and np.array_equal(a.indices, b.indices)


def test_main_rs_scene_file_equals_hard_coded_scene(pm):
    sd, st = pm.pbrt_scene.load_pbrt_file(os.path.join(HERE, "golden", "main_rs.pbrt"))
    assert st == {"integrator": "simplepath", "max_depth": 5, "spp": 256, "seed": 0, "sample_lights": True, "sample_bsdf": True}
    _same_scene(pm, sd, pm.scenes.main_rs_scene((400, 400)))


def test_directives_and_errors(pm):
    L = pm.pbrt_scene.load_pbrt
    head = 'LookAt 0 0 5  0 0 0  0 1 0\nCamera "perspective" "float fov" 40\nFilm "rgb" "integer xresolution" 32 "integer yresolution" 24\nWorldBegin\n'
    sd, st = L(head + '''
        LightSource "point" "point3 from" [0 3 0] "rgb I" [1 1 1] "float scale" 2
        Texture "grey" "spectrum" "constant" "rgb value" [.5 .5 .5]
        Material "diffuse" "texture reflectance" "grey"
        AttributeBegin
          Rotate 90 1 0 0
          Scale 2 2 2
          Shape "trianglemesh" "point3 P" [0 0 0  1 0 0  0 1 0  1 1 0] "integer indices" [0 1 2  1 3 2]
        AttributeEnd
        ReverseOrientation
        Shape "bilinearmesh" "point3 P" [-1 -1 0  1 -1 0  -1 1 0  1 1 0] "normal N" [0 0 1  0 0 1  0 0 1  0 0 1]
    ''')
    assert sd.film["full_resolution"] == (32, 24) and st["spp"] == 16
    assert len(sd.prims) == 2 and len(sd.prim_materials) == 3 and len(sd.materials) == 1
    tri, patch = sd.prims
    assert isinstance(tri, pm.TriangleMesh) and len(tri.indices) == 2
    # Rotate 90 about x then Scale 2: (1,1,0) -> (2, 0, 2) in world; render space = world - camera position (0,0,5)
    assert np.allclose(tri.positions[3], [2.0, 0.0, 2.0 - 5.0], atol=1e-5)
    assert patch.reverse_orientation and np.allclose(patch.normals, [[0, 0, -1]] * 4)
    assert sd.lights[0]["type"] == pm.host.PM_LIGHT_POINT and np.allclose(sd.lights[0]["m"][:3, 3], [0, 3, -5])
    for bad, msg in (('Shape "cylinder"', "not supported"), ('Material "conductor"', None), ('LightSource "spot"', "not supported"),
                     ('Texture "c" "spectrum" "checkerboard"', "constant"), ("MakeNamedMedium \"fog\"", "not supported"),
                     ("AttributeEnd", "without")):
        text = head + bad + ('\nShape "sphere"' if bad.startswith("Material") else "")
        with pytest.raises(pm.pbrt_scene.PbrtSceneError) as e:
            L(text)
        if msg:
            assert msg in str(e.value)
    with pytest.raises(pm.pbrt_scene.PbrtSceneError):
        L('WorldBegin')


@pytest.mark.gpu
def test_scene_file_renders_the_same_film(pm):
    sd, st = pm.pbrt_scene.load_pbrt_file(os.path.join(HERE, "golden", "main_rs.pbrt"))
    sd.film = pm.host.make_film(pm.host.Spectra(), (64, 48))
    sd.camera, _ = pm.host.make_camera(pm.host.Transform.look_at((3.0, 4.0, 1.5), (0.5, 0.5, 0.0), (0.0, 0.0, 1.0)).inverse(), 45.0, (64, 48))
    st["spp"] = 8
    a = pm.pbrt_scene.integrator_from(sd, st)
    a.render()
    b = pm.WavefrontIntegrator(5, True, True, pm.scenes.main_rs_scene((64, 48)), 8)
    b.render()
    assert np.array_equal(a.film().view(np.uint64), b.film().view(np.uint64))
