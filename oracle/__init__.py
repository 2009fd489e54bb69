"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the layer forward/backward hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline -- never as a compute path.
The product (``np_modeling_amd``) fails loudly when its HIP library is missing.
"""
