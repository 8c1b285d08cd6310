"""CPU oracle for the deeprl_network hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  It is a CPU restatement of the
reference's arithmetic (``/root/reference``: envs/cacc_env.py, agents/utils.py,
agents/policies.py, agents/models.py, utils.py) that the GPU path is checked
against.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package
``deeprl_network_amd`` never does and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * cacc_ref      -- PINNED bit-exactly (float64) against the real reference
                     ``CACCEnv`` run in the authoring container; vectors in
                     tests/golden/cacc_*.npz (made by tests/golden/make_golden_env.py).
  * nstep_ref     -- PINNED against the reference's OnPolicyBuffer /
                     MultiAgentOnPolicyBuffer (tests/golden/nstep_*.npz).
  * ortho_init    -- PINNED against agents/utils.py:ortho_init.
  * nn_ref        -- structure PINNED against the reference's own graph-building
                     code executed through oracle/tf1_shim (a fake TF1 API on
                     torch-CPU); TF-1.12 *kernel* semantics (RMSProp slots,
                     clip_by_global_norm, softmax) are restated, not executed:
                     "parity unpinned at the TF kernel boundary".
                     The heterogeneous (identical=False) nets are not restated
                     here: the product is compared directly with goldens produced
                     by the reference's own hetero code (nn_*_ragged.npz).
  * grid_ref      -- PARITY UNPINNED: SUMO is absent; the synthetic grid is
                     specified in this repo and grid_ref is its own oracle.
  * realnet_ref   -- topology / masks / widths PINNED against the reference's
                     real_net_env.py tables and map builders; dynamics PARITY
                     UNPINNED (SUMO + net file absent): own specification.
"""
