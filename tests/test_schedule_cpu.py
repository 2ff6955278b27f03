"""CPU test of the mirrored fusion scheduler, dispmap_super.binary_fuse_until_convergence
(dispmap_super.m:85-152), against schedules worked out by hand from the MATLAB source -- the
quirks included: the loop variable is bumped inside the body so ids(2) is fused first and ids(1)
only when the random tail names it (:116), `ids(diff(ids) == 0) = 0` drops the FIRST of a repeated
pair (:99), a proposal that did not change the energy is skipped until some other proposal does
(:118-122,135-148), and "changed" is the exact comparison E(end-1) ~= E(end) (:137).
No GPU: binary_fusion is replaced by a scripted energy model."""
import numpy as np
import pytest

from stereo_amd.dispmap import dispmap_super


class Scripted(dispmap_super):
    """binary_fusion(p) with integer proposals: the energy drops by drop[p] the first `times[p]` times."""

    def __init__(self, drops, times, e0=100.0):
        super().__init__([np.zeros((2, 2, 3)), np.zeros((2, 2, 3))], 1)
        self.stored_energy = e0
        self.drops, self.left = dict(drops), dict(times)
        self.fused = []

    def binary_fusion(self, p):
        self.fused.append(p)
        if self.left.get(p, 0) > 0:
            self.left[p] -= 1
            self.stored_energy = self.stored_energy - self.drops[p]
        return 0.0, 0.0, 0.0


def test_schedule_skips_ids1_and_stops_when_every_proposal_failed():
    dm = Scripted({1: 5.0, 2: 3.0, 3: 1.0}, {1: 1, 2: 1, 3: 1})
    dm.maxiter = 50
    # ids = [1 2 3 | 3 1 2 2 1 3 3 2 1] -> first of each repeated pair zeroed -> [1 2 3 1 2 1 3 2 1]
    n = dm.binary_fuse_until_convergence([1, 2, 3], rng=[3, 1, 2, 2, 1, 3, 3, 2, 1])
    assert dm.fused == [2, 3, 1, 2, 1, 3]      # ids(2) first; three improvements, then all three fail in turn
    assert n == 7                              # length(E): the start energy + six fusions
    assert dm.energy() == 100.0 - 9.0


def test_schedule_skips_marked_proposals_until_an_improvement():
    dm = Scripted({1: 5.0, 2: 3.0, 3: 1.0}, {1: 1, 2: 1, 3: 0})
    dm.maxiter = 50
    # ids = [1 2 3 | 2 3 2 1 3 2 1] (no repeats): 2 improves, 3 fails, 2 fails, 3 and 2 are skipped
    # (marked), 1 improves and clears the marks, 3 fails, 2 fails, 1 fails -> stop
    n = dm.binary_fuse_until_convergence([1, 2, 3], rng=[2, 3, 2, 1, 3, 2, 1])
    assert dm.fused == [2, 3, 2, 1, 3, 2, 1]
    assert n == 8


def test_schedule_respects_maxiter_and_exact_energy_compare():
    dm = Scripted({1: 1.0, 2: 1.0}, {1: 9, 2: 9})
    dm.maxiter = 2
    n = dm.binary_fuse_until_convergence([1, 2], rng=[1, 2, 1, 2, 1, 2, 1, 2, 1, 2])
    assert dm.fused == [2, 1] and n == 3       # maxiter loop trips, each fuses ids(iter + 1)
    # a change of one ulp is a change (E(end-1) ~= E(end) is exact): nothing is ever marked
    tiny = Scripted({1: np.spacing(100.0), 2: np.spacing(100.0)}, {1: 50, 2: 50})
    tiny.maxiter = 6
    n = tiny.binary_fuse_until_convergence([1, 2], rng=[1, 2] * 15)
    assert len(tiny.fused) == 6 and n == 7


def test_schedule_argument_check():
    from stereo_amd import StereoHipError
    dm = Scripted({}, {})
    with pytest.raises(StereoHipError, match="cell array"):
        dm.binary_fuse_until_convergence(np.zeros((4, 4)))
