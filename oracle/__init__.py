"""CPU oracle for the PWCLO-Net point-cloud operator path -- TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``pwclonet_pylidarslam_amd``) never
does; it has no CPU fallback and raises when its HIP library is missing.

``oracle.ops``    -- ctypes front-end of ``pointnet2_oracle.c`` (the nine
                     extension kernels + ``knn_point``), numpy in / numpy out.
``oracle.model``  -- pure-PyTorch CPU restatement of the PWCLO-Net layers built
                     on ``oracle.ops`` (validated against the imported
                     reference by ``oracle/gen_golden.py``).
"""
