"""The reference's own end-to-end check, restated: renders of `scenes/evaluation/*.json` against the reference-held images of
OTHER renderers (Mitsuba 2/3, Blender Cycles, Radiance) under `tests/golden/references/` (copied byte for byte from
/root/reference/scenes/evaluation/references by tests/golden/make_ref_images.py).

This is what turns "oracle == HIP" (two restatements by the same hand) into "agrees with what the reference is validated
against": a misreading of the Artic sources shared by both would show up here.

  * CPU (`-m "not gpu"`): the oracle at 256 x 256, 64 spp. Mean radiance within 1.5 % of the reference image and the
    reference's error metric (`error_image`, scripts/RunEvaluations.py:86-95) on 8 x 8 box-filtered images (64 spp x 64
    pixels per cell ~ the noise of 4096 spp) below 2e-3, per-scene exceptions stated with their reason.
  * GPU (`-m gpu`): the HIP path at 256 x 256, 1024 spp — the sample count RunEvaluations.py defaults to — judged exactly as
    the reference judges itself: `error_image` at full resolution below `predef_eps[scene]` or 1e-3 (RunEvaluations.py:98-123,
    164,181-195), plus the 1 % mean bound of MakeHtml-style comparisons.

Scenes whose reference image cannot pin anything are listed in EXCLUDED with the triage.
"""
import os
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, SCENES

EVAL = os.path.join(SCENES, "evaluation")
REFS = os.path.join(GOLDEN, "references")

# scripts/RunEvaluations.py:98-123
PREDEF_EPS = {"two-planes-mirror": 2e-2,  # (not in the reference's table: the mirror caustic, see PINNED)
              "two-planes-plastic": 2e-3,  # (not in the table either: 1.2e-3 at 1024 spp against a Radiance image with visible ambient-cache blotches)
              "two-planes-brtdfunc1": 2e-3, "two-planes-brtdfunc2": 2e-3, "two-planes-brtdfunc3": 2e-3,  # (as two-planes-plastic: the same Radiance setup)
              "cbox-d1": 5e-3, "cbox-d6": 5e-3, "cycles-lights": 5e-2, "cycles-principled": 5e-2, "cycles-tex": 1e-2, "cycles-sun": 1e-2,
              "cycles-mix-diff-trans": 5e-3, "room": 1e-3, "volume": 5e-3, "env4k": 2e-3, "multilight-uniform": 3e-4, "multilight-simple": 3e-4,
              "multilight-hierarchy": 3e-4, "sphere-light-ico": 2e-3, "sphere-light-ico-nopt": 2e-3, "sphere-light-uv": 2e-3, "sphere-light-pure": 3e-3}
DEFAULT_EPS = 1e-3

# scene -> (mean tolerance, bound on the 8x8-filtered error at 64 spp); None = defaults (0.015, 2e-3)
PINNED = {
    "cbox-d1": None,                                      # Mitsuba: plane area light (Urena), depth 1
    "cbox-d6": (0.025, 2e-3),                             # ... with diffuse interreflection, depth 6: +1.6 % away from the lamp (+0.9 % overall); the
                                                          # reference's own bound for the two cbox scenes is 5e-3 instead of 1e-3
    "cycles-box": None,                                   # Cycles: area light + diffuse box
    "cycles-mix-diff-diff": None,                         # Cycles: blend of two diffuse BSDFs
    "cycles-mix-diff-trans": None, "cycles-mix-trans-trans": None,  # Cycles: blends with tinted transparent BSDFs (delta inside a mix)
    "cycles-tex": (0.035, 2e-3),                          # Cycles: principled BSDF with a JPEG base-colour texture, point light + environment: 2.4 % darker
                                                          # than Cycles (the reference's own bound for this scene is 1e-2)
    "cycles-sun": (0.015, 8e-3),                          # Cycles: sun (cone) light; the penumbra differs slightly (reference's own eps: 1e-2)
    "emissive-plane": None, "emissive-plane-nopt": None,  # Mitsuba: emissive-hit MIS, plane and mesh-area ("optimize": false) samplers
    "emissive-plane-scale": None, "emissive-plane-scale-nopt": None,
    "multilight": None, "multilight-uniform": None, "multilight-hierarchy": None, "multilight-simple": None,  # Mitsuba: many lights; uniform / hierarchy / flux-CDF selectors
    "plane-array-diffuse": None,                          # Radiance: sun + sky over diffuse planes
    "plane-d1": None, "plane-d6": None,                   # Mitsuba: environment + plane
    "point": None,                                        # Mitsuba: point light
    "room": None,                                         # Mitsuba: room lit through a window
    "sky-clear": None, "sky-cloudy": None, "sky-intermediate": None, "sky-uniform": None,  # Radiance gensky: the four CIE models
    "sky-perez1": None, "sky-perez2": None, "sky-perez3": None,  # Radiance gendaylit -P: Perez all-weather sky with its sun (clear, overcast, intermediate)
    "sphere-light-ico": None, "sphere-light-ico-nopt": None,  # Mitsuba: sphere light; the icosphere mesh is recognised as a sphere (getAsSphere) / sampled as a mesh
    "sphere-light-pure": None,                            # Mitsuba: the analytic sphere shape with the sphere emitter
    "sphere-light-uv": (0.03, 3e-3),                      # the same with a coarse uv-sphere: the mesh has ~2 % less area than the sphere
    "sun-on-plane": None,                                 # Radiance: sun over a plane
    "volume": (0.045, 3e-3),                              # Mitsuba: an absorbing sphere in a room (volumetric path tracer). A uniform +3.3 % over the
                                                          # whole image, walls included; the reference's own bound for this scene is 5e-3, i.e. it
                                                          # expects a bias of this size itself (at 1e-3 a 3.3 % offset alone would fail)
    "two-planes-brtdfunc1": None, "two-planes-brtdfunc2": None, "two-planes-brtdfunc3": None,  # Radiance BRTDfunc with constant arguments: mirror +
                                                          # perfect transmission + diffuse reflection in three mixtures (make_add_bsdf, bsdf/rad.art:7-29)
    "two-planes-plastic": None,                           # Radiance: a 1 cm sphere light (analytic sphere) over two diffuse planes
    "two-planes-mirror": (0.015, 2.5e-2),                 # Radiance: the same with a mirror; the caustic the mirror throws on the floor reaches a
                                                          # path tracer only through BSDF-sampled hits of the 1 cm emitter (fireflies; Radiance
                                                          # mirrors the light as a virtual source), the rest of the image agrees (median ratio 1.000)
    # Cycles: a metallic principled sphere whose base colour is an expression over a bitmap, on an expression checkerboard floor, lit by a
    # point light; -normalmap / -bumpmap wrap it in a "transform" BSDF whose normal is an expression (texture-space normal map; bump() over
    # luminance differences). Judged by LIT_MEDIAN below instead of error_image, with the mean bound given here.
    "cycles-roughness-rxry": (0.03, None), "cycles-roughness-raniso": (0.03, None), "cycles-normalmap": (0.035, None), "cycles-bumpmap": (0.03, None),
    "cycles-lights": (0.25, 5e-2),                        # Cycles: point + spot + area light given as Blender watts; the reference's own bound
                                                          # for this scene is 5e-2 ("should resemble Cycles, no need to be exact")
}

# Scenes judged by the median ratio over the lit part of the image (reference > 0.05) instead of error_image: the blue squares of the
# sphere's texture have red = 0 and green = 0.002, and the floor next to the sphere is lit only through glossy reflections of the point
# light (which Cycles filters), so the relative metric is decided by channels and pixels that are ~0 in the reference (0.03 - 0.4 at any
# sample count) while the lit image agrees to a few tenths of a percent (medians 1.0001 / 1.0009 / 1.0043 at 16 384 spp on the GPU).
LIT_MEDIAN = {"cycles-roughness-rxry": 0.01, "cycles-roughness-raniso": 0.01, "cycles-normalmap": 0.01, "cycles-bumpmap": 0.015}


def lit_median_ratio(img, ref):
    a, b = img.mean(axis=2), ref.mean(axis=2)
    lit = b > 0.05
    assert lit.sum() >= 256
    return float(np.median(a[lit] / b[lit]))


EXCLUDED = {
    "three-planes-brtdfunc1": "as three-planes-glass: the light sits behind the BRTDfunc pane, whose specular transmission lets no shadow ray through "
                              "(a delta lobe inside a non-delta BSDF), so the floor in front of the pane gets its direct light only from BSDF-sampled hits of "
                              "the 1 cm emitter; mean 0.94 x the Radiance image, error_image 2.5e-2 on 8 x 8 cells at 64 spp. The BSDF itself is pinned by "
                              "two-planes-brtdfunc1..3 (means within 0.8 %, error_image < 1e-3).",
    "three-planes-roos": "the same geometry with the Roos glazing model; in addition RadRoosBSDF.cpp hands its refl_* properties to the trns_* "
                         "arguments of make_rad_roos_bsdf and vice versa (bsdf/rad.art:36-39), restated as written: the pane reflects what it should "
                         "transmit (mean 1.19 x the Radiance image).",
    "cycles-lights-lt": "the light tracer as written is not normalised to the path tracer: camera connections are weighted with image_area = 1 "
                        "instead of the pixel's importance (src/artic/camera/perspective.art:36,47-51), so the image is the direct + indirect lighting "
                        "times the area of the image plane at distance 1 (4 sx sy), and a spot light's emission carries another 1 / spot_area "
                        "(light/spot.art:41-47). Both relations are asserted against the path tracer in tests/test_lighttracer.py instead.",
    "cycles-lights-ppm": "the photon mapper starts its light paths with Light::sample_emission, and the spot light's carries a factor 1 / spot_area = "
                         "1 / (pi tan^2 cutoff) that its sample_direct does not (src/artic/light/spot.art:13-14,41-47; the same factor the light tracer "
                         "shows, cycles-lights-lt): the green channel — the spot light — is 0.54 x the path tracer's and the Cycles image, red (area light) and "
                         "blue (point light) are 1.26 / 0.99 x Cycles exactly like the path tracer's cycles-lights (1.27 / 1.00). "
                         "tests/test_photonmapper.py asserts green x spot_area == path tracer's green within 2 %, and — with an oracle-only switch — that "
                         "storing photon directions through a sound 16-bit encoding instead of encode_signed_norm_16 as written (core/common.art:186-197) "
                         "does not change the image (round 3 had blamed that encoding).",
    "env": "make_environment_light_textured.sample_dir (src/artic/light/env.art:112-113) returns tex(ctx) without `scale`, emission "
           "(:139-144) multiplies by it: with scale 100 the NEE half of the estimator is 100x too dark. Restated as written (bug-compatible), "
           "so the image cannot match Mitsuba's; with the scale folded into the texture NEE and BSDF-only sampling agree with each other "
           "and sit a uniform 2.18x above the Mitsuba image (single bright texel: the texel -> direction convention is not pinned).",
    "flipped-prim-diffuse": "background exact (0.8); the cylinder's unlit side is 0.640 here = albedo 0.8 x environment 0.8, the value a convex "
                            "Lambertian body must show, while the Cycles image has 0.50 on body AND cap (i.e. an effective albedo of ~0.62: the "
                            "JSON was edited by hand after the export, scenes/evaluation/README.md); the lit side is another factor ~pi up "
                            "because the exporter writes Blender watts as Ignis `power` (scripts/blender_exporter/ignis_blender/light.py:57-66).",
    "flipped-prim-glass": "the same base scene as flipped-prim-diffuse with a rough-glass cylinder: the background is exact (ratio 1.00 along the "
                          "image border), the cylinder is lit by the same 1000 W point light whose Blender-watt convention makes an Ignis render ~pi "
                          "brighter than the Cycles image, and what the point light sends through the rough glass arrives as fireflies (8 x 8 blocks up to "
                          "60x at 128 spp).",
    "sun-on-plane-and-stick": "the sun of the JSON / .rad sits exactly on the horizon (direction z = 0) and lights the stick from the right; the "
                              "Radiance image shows an evenly lit plane without a cast shadow and the stick lit from the left: image and scene disagree.",
    "three-planes-dielectric": "the camera looks through a single dielectric interface (ior 1.55) at a plane and a light INSIDE the medium. The image "
                               "is a uniform 2.42x the Radiance one = n^2 (2.4025): make_pure_dielectric_bsdf scales a refracted sample by k^2 only "
                               "for adjoint paths (src/artic/bsdf/dielectric.art:27-29), i.e. camera paths carry no 1/n^2 radiance scaling; Radiance applies it.",
    "three-planes-interface": "as three-planes-dielectric with ior 1.55 against 1.22 (2.1x).",
    "three-planes-glass": "a thin glass pane between the 1 cm sphere light and the floor: shadow rays stop at the pane, so the floor in front is lit only by "
                          "BSDF-sampled paths that happen to hit the emitter (pure noise at any feasible sample count; Radiance lets shadow rays through "
                          "`glass`). The part seen through the pane agrees (ratio 1.03).",
}


def reference_for(stem):
    """get_reference_path (scripts/RunEvaluations.py:16-36): shortest ref-<name>*.exr, dropping '-' sections if there is none."""
    base = stem
    for _ in range(3):
        found = [str(p) for p in Path(REFS).glob(f"ref-{base}*.exr")]
        if found:
            return min(found, key=len)
        if "-" not in base:
            break
        base = base[:base.rfind("-")]
    return None


def error_image(img, ref):
    """scripts/RunEvaluations.py:86-95: relative squared error where the reference is non-zero, absolute squared error
    elsewhere, clipped at the 99th percentile, averaged."""
    mask = ref != 0
    err = np.zeros_like(ref)
    err[mask] = np.square((img[mask] - ref[mask]) / ref[mask])
    err[~mask] = np.square(img[~mask])
    top = np.percentile(err, 99)
    return float(np.average(np.clip(err, 0, top)))


def robust_mean_ratio(img, ref):
    """Ratio of the means with both images clipped at the reference's 99.5th percentile: a handful of pixels showing a 1 cm
    emitter of radiance 10 000 would otherwise decide the mean (the reference's own metric clips at a percentile as well)."""
    top = np.percentile(ref, 99.5)
    return float(np.minimum(img, top).mean() / np.minimum(ref, top).mean())


def box(img, f):
    h, w, c = img.shape
    return img.reshape(h // f, f, w // f, f, c).mean(axis=(1, 3))


def _load(stem):
    import sys
    sys.path.insert(0, GOLDEN)
    import exr_decode
    from ignis_amd.tables import LoadedScene
    ref = exr_decode.read_rgb(reference_for(stem))
    return LoadedScene.from_file(os.path.join(EVAL, stem + ".json"), 256, 256), ref


def test_every_usable_reference_image_is_listed():
    """Of the evaluation scenes the loader accepts, each one with a reference image is either pinned or excluded in writing."""
    from ignis_amd.tables import LoadedScene
    loadable = []
    for p in sorted(Path(EVAL).glob("*.json")):
        if "-base" in p.stem or reference_for(p.stem) is None:
            continue
        try:
            LoadedScene.from_file(str(p), 32, 32)
            loadable.append(p.stem)
        except RuntimeError:
            pass
    assert sorted(loadable) == sorted(list(PINNED) + list(EXCLUDED))
    assert len({reference_for(s) for s in PINNED}) >= 10  # distinct reference images


@pytest.mark.parametrize("stem", sorted(PINNED))
def test_oracle_matches_reference_image(stem):
    import oracle
    scene, ref = _load(stem)
    fb = np.zeros((256, 256, 3), np.float32)
    for it in range(4):
        f, _ = oracle.render(scene, 16, 256, 256, iteration=it, seed=1)
        fb += f
    fb /= 4
    assert np.isfinite(fb).all()
    mean_tol, err_tol = PINNED[stem] or (0.015, 2e-3)
    ratio = robust_mean_ratio(fb, ref)
    assert abs(ratio - 1) <= mean_tol, f"{stem}: mean radiance {ratio:.4f} x the reference image"
    if stem in LIT_MEDIAN:
        med = lit_median_ratio(box(fb, 8), box(ref, 8))
        assert abs(med - 1) <= LIT_MEDIAN[stem], f"{stem}: median ratio over the lit 8x8 cells {med:.4f}"
        return
    err = error_image(box(fb, 8), box(ref, 8))
    assert err <= err_tol, f"{stem}: error_image on 8x8-filtered images {err:.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("stem", sorted(PINNED))
def test_hip_matches_reference_image_as_the_reference_judges_itself(gpu_device, stem):
    scene, ref = _load(stem)
    gpu_device.assign_scene(scene)
    gpu_device.resize(256, 256)
    gpu_device.clear_framebuffer()
    spi, iterations = 16, 64  # 1024 spp (RunEvaluations.py --spp default)
    for it in range(iterations):
        gpu_device.render(spi, 256, 256, iteration=it, seed=1)
    fb = gpu_device.framebuffer() / iterations
    assert np.isfinite(fb).all()
    err = error_image(fb, ref)
    eps = PREDEF_EPS.get(stem, DEFAULT_EPS)
    ratio = robust_mean_ratio(fb, ref)
    mean_tol = (PINNED[stem] or (0.015, 0))[0]
    print(f"{stem}: error_image {err:.3e} (eps {eps:g}), mean ratio {ratio:.4f}")
    if stem in LIT_MEDIAN:
        med = lit_median_ratio(fb, ref)
        assert abs(med - 1) <= LIT_MEDIAN[stem] and abs(ratio - 1) <= mean_tol, f"{stem}: median ratio over the lit pixels {med:.4f}, mean ratio {ratio:.4f}"
        return
    assert err < eps, f"{stem}: error_image {err:.3e} >= {eps:g} (the reference's own pass criterion)"
    assert abs(ratio - 1) <= (0.01 if PINNED[stem] is None else mean_tol), f"{stem}: mean radiance {ratio:.4f} x the reference image"
