"""Global bundle adjustment (reference: eff_impl=True, dpvo/fastba/block_e.cu:43-300, dpvo.py:312-326).

Routed here when `eff_impl=True` or when more than 20 poses are free (6N > 120, beyond the dense in-LDS Schur path
of dpvo_amd/csrc/ba.hip).  The block-sparse device path (E_lookup blocks + rocSOLVER potrf/potrs for 6N up to ~6000)
is the next SURVEY.md section-8 row to build (BASELINE config 5, LOOP_CLOSURE=True); it is NOT implemented yet and
fails loudly instead of falling back to anything else.
"""
from .. import _lib as L


def global_BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M, iterations):
    raise L.DPVOHipError(
        f"global BA with {t1 - t0} free poses (eff_impl) is not implemented yet in dpvo_amd "
        "(dense Schur path covers t1 - t0 <= 20); LOOP_CLOSURE=True configurations are the next scope row")
