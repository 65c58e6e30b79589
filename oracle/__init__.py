"""CPU oracle for the CM-UNet hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path
(``cmunet_amd``) never imports this package and fails loudly
when its HIP library is missing.

The reference's arithmetic for this path is floating point and lives in
PyTorch ATen ops, so the restatement is written with
plain ``torch`` CPU functional ops in fp32/fp64 -- no ``nn.Module`` from the
reference, no code copied from it.  Each function cites the reference
file:line it follows.

Pinning (see DESIGN.md "Oracle"):
  * dense UNet blocks and tensor losses are pinned against the reference
    itself, imported in the build container by ``oracle/gen_golden.py``
    (the reference has no tests / golden vectors of its own: SURVEY F2);
  * the CM-UNet (cmae.*) and MoCo (moco2_module.py) parts need mmengine / mmcv /
    pytorch-lightning / torchvision, which this image lacks, and hard-code the
    device: ``gen_golden.py`` runs the reference's own modules behind plumbing
    stand-ins for those libraries (registry, base classes, one-rank all_gather,
    SyncBN -> BatchNorm1d, a CPU redirect of .cuda(); the same policy as the timm
    stand-in of the SparK fixtures) and writes ``tests/golden/cmunet_ref.npz`` /
    ``moco_ref.npz`` after asserting oracle == reference on every number.  What
    only exists on more than one rank (label offsets B*rank, gathered keys,
    shuffle-BN) stays restated and is covered by the gloo tests.
"""
