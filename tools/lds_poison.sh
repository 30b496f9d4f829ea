#!/bin/bash
# Run the GPU parity suite on an experiment build whose feature kernels start with NaN-filled LDS (-DHIPFEAT_LDS_POISON):
# any read of an LDS location the kernel never wrote turns into a NaN in the output and fails a test.
# usage (GPU box): python tools/variants.py poison:"-DHIPFEAT_LDS_POISON" (here) ; tools/lds_poison.sh (there)
HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_poison.so python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_layout_launch_is_hip_graph_capturable "$@"  # (the poison build sets a device symbol per launch: not capturable)
