"""BASELINE.json config 4: PLOC++ (and the other builders) -> LBVH-layout adapter -> while-while traversal -> image, pixel-exact
against the CPU oracle's traversal of the oracle's tree; plus the same rays/tree through the REFERENCE's own GenerateRays /
BvhTraversalWhile kernels when oracle/_ref is built."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
W = 512          # the reference's image size (src/TwoPassLbvh.cpp:219-220)


def scenes(pkg):
    mg = pkg.meshgen
    cam, xf = pkg.cornell_view()
    out = {"cornell382": (mg.load_tri(os.path.join(GOLDEN, "cornell382.tri")), cam, xf)}
    # sponza-like room 30x12x18 seen from inside
    cam2 = cam.copy(); xf2 = xf.copy()
    cam2["eye"][0] = (15.0, 6.0, 17.0, 0.0); cam2["quat"][0] = pkg.qt_rotation((0.0, 1.0, 0.0, 0.0)); xf2["translation"][0] = (0.0, 0.0, 0.0)
    out["sponza_40k"] = (mg.sponza_like(40_000, 3), cam2, xf2)
    cam3 = cam.copy(); xf3 = xf.copy()
    cam3["eye"][0] = (0.5, 0.5, 2.2, 0.0); cam3["quat"][0] = pkg.qt_rotation((0.0, 1.0, 0.0, 0.0)); xf3["translation"][0] = (0.0, 0.0, 0.0)
    out["uniform_20k"] = (mg.uniform(20_000, 31), cam3, xf3)
    return out


@pytest.fixture(scope="module")
def views(pkg):
    return scenes(pkg)


@pytest.mark.parametrize("scene", ["cornell382", "sponza_40k", "uniform_20k"])
@pytest.mark.parametrize("algo", [2, 1, 3])
def test_image_pixel_exact_vs_oracle(pkg, orc, ctx, views, scene, algo):
    tris, cam, xf = views[scene]; n = len(tris)
    b = pkg.BUILDERS[algo]().build(ctx, tris)
    rgba, rays = b.render(tris, cam, xf, W)
    assert rgba[3::4].sum() > 255 * 1000, "the view must actually see geometry"
    ref = orc.build_tree(algo, tris)
    onodes = ref["nodes"] if ref["layout"] == 0 else orc.ploc_to_lbvh_layout(ref["nodes"], ref["leaves"])
    img, overflow = orc.trace_while(rays, tris, onodes, xf, ref["root"], W, n - 1)       # same rays (tanf differs between libm and OCML)
    assert overflow == 0, "scene exceeds the reference's 32-entry traversal stack; pick another view"
    assert np.array_equal(rgba, img), f"{np.count_nonzero(rgba != img)} of {rgba.size} bytes differ"
    # rays: device tanf vs libm tanf may differ by an ulp; everything else is IEEE-exact
    orays = orc.generate_rays(cam, W, W)
    assert np.allclose(rays["direction"], orays["direction"], atol=2e-6) and np.array_equal(rays["origin"], orays["origin"])


@pytest.mark.parametrize("scene", ["cornell382", "sponza_40k"])
def test_image_vs_reference_kernels(pkg, orc, ctx, views, scene):
    if not os.path.exists(orc.REF_DRIVER):
        pytest.skip("oracle/_ref not built")
    tris, cam, xf = views[scene]; n = len(tris)
    b = pkg.PLOCNew().build(ctx, tris)
    rgba, rays = b.render(tris, cam, xf, W)
    nodes = b.to_lbvh_layout()
    # reference GenerateRays built without FP contraction == product rays, bit for bit
    assert orc.ref_generate_rays(cam, W, W, nofma=True).tobytes() == rays.tobytes()
    # reference BvhTraversalWhile on the product's tree and rays: contraction off -> pixel exact; hiprtc defaults -> at most a few
    # +-1 pixel values (rounding of u*255 under FMA contraction)
    img_nofma = orc.ref_trace_while(rays, tris, nodes, xf, 0, W, n - 1, nofma=True)
    assert np.array_equal(rgba, img_nofma), f"{np.count_nonzero(rgba != img_nofma)} bytes differ vs reference kernel (contract off)"
    img_default = orc.ref_trace_while(rays, tris, nodes, xf, 0, W, n - 1, nofma=False)
    diff = rgba.astype(np.int16) - img_default.astype(np.int16)
    assert np.count_nonzero(diff) <= rgba.size // 1000 and np.abs(diff).max() <= 1 or np.count_nonzero(diff) == 0
