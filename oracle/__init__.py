"""CPU oracle for the rasterizer hot path — TEST INFRASTRUCTURE ONLY (parity unpinned,
see oracle/c/sgn_oracle.c and oracle/torch_oracle.py headers)."""
