"""INTEGRATION.md variant A compiles: the reference's call site (controllers: SiftFeatureMatcher(options, &database,
&cache); Setup(); cache.Setup(); Match(pairs) with colmap::Database* / colmap::FeatureMatcherCache*) against
dagsfm_amd/host/colmap_traits.h, over headers that carry the reference's exact signatures (tests/colmap_stub --
Eigen and the reference's other dependencies are not installed here).  VERDICT r01 item 9."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_call_site_compiles_against_colmap_traits(tmp_path):
    stub = os.path.join(ROOT, "tests", "colmap_stub")
    obj = str(tmp_path / "call_site.o")
    # the reference's own flags (/root/reference/src/CMakeLists.txt:37: -std=c++11 -Wall), warnings and extensions as errors
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-pedantic-errors", "-c", os.path.join(stub, "call_site.cc"),
                           "-I", stub, "-I", ROOT, "-o", obj])
    # the object refers to the C-ABI entry points and to nothing of this repository's own host types
    syms = subprocess.check_output(["nm", "-C", "--undefined-only", obj]).decode()
    for s in ("dsm_ctx_create", "dsm_set_images", "dsm_match_pairs", "dsm_verify_pairs", "dsm_get_two_view_geometries",
              "colmap::FeatureMatcherCache::WriteTwoViewGeometry", "colmap::Database::ImagePairToPairId"):
        assert s in syms, s
    assert "dagsfm_amd::Database" not in syms and "dagsfm_amd::FeatureMatcherCache" not in syms
