"""csrc/verify_roots_refill.h carries pr_hessenberg_eigenvalues' and pr_poly_roots' own lines (init / one pass of the loop / finish; begin /
end): regenerated here and compared with the committed header, so an edit of verify_linalg.h cannot leave the experimental refill kernels
on an older algorithm."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_refill_header_is_the_generators_output():
    import gen_roots_refill
    committed = open(os.path.join(ROOT, "dagsfm_amd", "csrc", "verify_roots_refill.h")).read()
    assert gen_roots_refill.generate() == committed
