"""The scene ingestion (pbrt-rs_amd/pbrt_scene.py) against the known-answer vectors of the reference's OWN
parser tests (tests/golden/reference_parser_vectors.json: inputs and expected outcomes of scene_parsing/common.rs:620-716,
scene.rs:282-351, directives/lights.rs:96-114, directives/cameras.rs:160-177).  The reference's parser stops at todo!() after
parsing (construct.rs:57), so these vectors pin the SYNTACTIC layer: literals, typed parameters, the parameter map, comments,
unknown and repeated directives."""
import importlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ps():
    return importlib.import_module("pbrt-rs_amd").pbrt_scene


@pytest.fixture(scope="module")
def vec():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_parser_vectors.json")))


def _value(tok):
    """a literal token as the value the reference's AtomicLiteral holds"""
    if tok.startswith('"'):
        return tok.strip('"')
    if tok in ("true", "false"):
        return tok == "true"
    return float(np.float32(tok))


def _one_param(ps, text):
    (_, _, params), = ps._parse("X " + text)
    (name, (typ, vals)), = params.items()
    return name, typ, [_value(v) for v in vals]


def test_atomic_literals(ps, vec):
    for text, want in vec["atomic_literals"]:
        _, _, vals = _one_param(ps, f'"t v" {text}')
        assert len(vals) == 1
        if isinstance(want, float):
            assert vals[0] == pytest.approx(want, rel=1e-7) and not isinstance(vals[0], bool)
        else:
            assert vals[0] == want and type(vals[0]) is type(want)
    for text in vec["atomic_literal_errors"]:
        with pytest.raises(ps.PbrtSceneError):
            ps._parse(f'X "t v" {text}')


def test_array_literals(ps, vec):
    text, want = vec["array_literal"]
    _, _, vals = _one_param(ps, f'"float v" {text}')
    assert vals == pytest.approx(want, rel=1e-7)
    for text in vec["array_literal_errors"]:
        with pytest.raises(ps.PbrtSceneError):
            ps._parse(f'X "float v" {text}')


def test_params(ps, vec):
    for key in ("param", "rgb_param"):
        text, name, typ, want = vec[key]
        got = _one_param(ps, text)
        assert got[0] == name and got[1] == typ and got[2] == pytest.approx(want, rel=1e-7)
    with pytest.raises(ps.PbrtSceneError):
        ps._parse("X " + vec["rgb_param_wrong_length"])
    text, want = vec["param_map"]
    (_, _, params), = ps._parse("X " + text)
    assert set(params) == set(want)
    for name, (typ, vals) in want.items():
        assert params[name][0] == typ and [_value(v) for v in params[name][1]] == pytest.approx(vals)


def test_entity_directives(ps, vec):
    text, want = vec["distant_light"]
    (name, args, params), = ps._parse(text)
    assert name == "LightSource" and args == ["distant"]
    assert params["L"][0] == "rgb" and [_value(v) for v in params["L"][1]] == pytest.approx(want["L"], rel=1e-7)
    assert params["from"][0] == "point" and [_value(v) for v in params["from"][1]] == pytest.approx(want["from"], rel=1e-7)
    text, want = vec["orthographic_camera"]
    (name, args, params), = ps._parse(text)
    assert name == "Camera" and args == ["orthographic"]
    for k, v in want.items():
        assert params[k][0] == "float" and _value(params[k][1][0]) == pytest.approx(v, rel=1e-7)


def test_comments_are_stripped(ps, vec):
    with_comments, without = vec["strip_comments"]
    assert list(ps._tokens(with_comments)) == list(ps._tokens(without))
    assert ps._parse(with_comments) == ps._parse(without)
    assert [d[0] for d in ps._parse(with_comments)] == ["LookAt", "Camera"]


def test_unknown_and_repeated_directives_are_errors(ps, vec):
    """scene.rs test_invalid_directive_name / test_global_options_fail_on_repeat_directive.  (The orthographic camera of these
    vectors parses upstream but cannot render there - OrthographicCamera::generate_ray is todo!() - so the loader is given the
    same text with the perspective camera.)"""
    persp = lambda t: t.replace('"orthographic" "float shutteropen" 1.2 "float shutterclose" 2.4', '"perspective" "float fov" 45')
    with pytest.raises(ps.PbrtSceneError, match="ThisIsNotARealDirective"):
        ps.load_pbrt(persp(vec["invalid_directive"]))
    with pytest.raises(ps.PbrtSceneError, match="repeated directive Camera"):
        ps.load_pbrt(persp(vec["repeat_directive"]))
    sd, settings = ps.load_pbrt(persp(vec["scene_ok"]))
    assert len(sd.prims) == 1 and sd.prims[0].radius == np.float32(0.25)
    with pytest.raises(ps.PbrtSceneError, match="orthographic"):
        ps.load_pbrt(vec["scene_ok"])
