"""BASELINE.json config 4: PLOC++ (and the other builders) -> LBVH-layout adapter -> while-while traversal -> image, pixel-exact
against the CPU oracle's traversal of the oracle's tree; plus the same rays/tree through the REFERENCE's own GenerateRays /
BvhTraversalWhile kernels when oracle/_ref is built."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, require_ref

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
    require_ref(os.path.exists(orc.REF_DRIVER), "oracle/_ref/libref_driver.so (the reference's kernels)")
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


@pytest.mark.parametrize("scene", ["cornell382", "sponza_40k", "uniform_20k"])
@pytest.mark.parametrize("algo", [2, 3])
def test_other_traversal_flavours(pkg, orc, ctx, views, scene, algo):
    """restart trail / if-if / speculative while-while (bvh_trace): pixel-exact and triangle-test-count-exact against the reference's own
    kernels on the same tree and rays (contraction off).  if-if visits in the while-while order, so it also renders the while-while
    image everywhere; the restart trail enters the LEFT box first when both are entered at the same distance (while-while: the right
    one) and the speculative kernel prunes with a hit distance that lags by one leaf, so where triangles tie exactly (the Sponza-like
    room's coincident wall quads) they — the reference's kernels and these alike — keep a different one of the tied triangles."""
    tris, cam, xf = views[scene]; n = len(tris)
    b = pkg.BUILDERS[algo]().build(ctx, tris)            # PLOC layouts: root 0, which the restart trail requires
    base, rays = b.render(tris, cam, xf, W)
    nodes = b.to_lbvh_layout()
    have_ref = os.path.exists(orc.REF_DRIVER)
    for kind in (1, 2, 3):
        img, _, cnt = b.render(tris, cam, xf, W, kind=kind, counts=True)
        assert cnt.max() > 0 and img[3::4].sum() == base[3::4].sum(), "same coverage"
        if kind == 2 or scene != "sponza_40k":
            assert np.array_equal(img, base), f"kind {kind}: {np.count_nonzero(img != base)} bytes differ from the while-while image"
        if have_ref:
            rimg, rcnt = orc.ref_trace_kind(kind, rays, tris, nodes, xf, 0, W, n - 1, nofma=True)
            assert np.array_equal(img, rimg), f"kind {kind}: {np.count_nonzero(img != rimg)} bytes differ vs the reference kernel"
            if kind in (1, 2):
                assert np.array_equal(cnt, rcnt), f"kind {kind}: triangle-test counts differ in {np.count_nonzero(cnt != rcnt)} rays"
    # argument validation: the restart trail needs root 0
    lbvh = pkg.SinglePassLbvh().build(ctx, tris)
    if lbvh.result.root != 0:
        with pytest.raises(pkg.BvhError):
            lbvh.render(tris, cam, xf, W, kind=1)
