"""CPU oracle for the VLMo pretraining hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain-PyTorch fp32 restatement of the
reference algorithm (fanzhongyi/ExploreMultiModal, ``models/vlmo/vlmo.py``,
``models/vlmo/vlmo_module.py``, ``dall_e/encoder.py``) plus the deterministic
synthetic-weight / synthetic-batch recipes used to build golden vectors.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / reported CPU baseline.  The product
package ``exploremultimodal_amd`` never imports it and has no CPU fallback: it
raises if the HIP library is missing.

Parity pin: the reference holds no tests or golden vectors, so
the oracle is pinned by outputs of the reference itself, executed in the build
container by ``oracle/gen_golden.py`` and committed under ``tests/golden/``.
"""
