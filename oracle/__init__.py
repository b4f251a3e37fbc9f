"""CPU oracle for the batched-decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
engine: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.  The product path
(``vllm-backend-public_amd/mi355_decode``) never imports this package and
fails loudly when its HIP library is missing.

Pinning status
--------------
* Arithmetic ops (``ops_ref``, ``model_ref``): the reference repository holds
  no implementation and no test vector for them -- the arithmetic lives in the
  ``vllm/vllm-openai:v0.9.1`` image it launches
  (benchmark-vllm/vllm_manager.py:371-375), which is not installed here.
  **Parity with stock vLLM is therefore unpinned.**  The restatement is pinned
  instead against HF ``transformers`` (5.15.0, CPU, fp32) run in the build
  container; the vectors are committed under ``tests/golden/`` together with
  the generating script ``oracle/gen_golden.py``.
* Harness contracts (``harness_ref``): pinned against the reference's own
  importable modules (benchmark-vllm/models.py, vllm_templates.py,
  benchmark-results/models.py) and its README known answers; see
  ``tests/golden/harness_*.json``.
"""
