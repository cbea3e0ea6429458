"""CPU oracle for the MuZero self-play search path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``muzero_amd``) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and there only as the checker /
the timed CPU baseline, never as a fallback for the HIP path.

It restates, on the CPU, the algorithm of the reference's ``model.py`` /
``train.py`` for the search hot path:

* ``oracle.tree``  - per-node search tree: PUCT score, child selection,
  expansion, Dirichlet mixing, discounted backup with min-max statistics,
  the wave-W ``run`` and the final action choice (model.py:351-546).
* ``oracle.nets``  - torch fp32 mirror of the representation / dynamics /
  prediction networks and the value-support decode (model.py:89-109,
  143-348); parameter names match the reference ``state_dict``.
* ``oracle.prep``  - frame stack and observation preparation
  (train.py:22-63, train.py:529-540).

Parity pin: every function here is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were generated in the build
container by importing the reference's ``model.py`` (generator:
``tests/golden/make_golden.py``).  ``oracle.prep`` restates ``train.py``, which
cannot be imported (gymnasium / ale_py absent); it is pinned by an exhaustive
256-value check of the normalisation arithmetic and by the ordering rule read
from the source.
"""
