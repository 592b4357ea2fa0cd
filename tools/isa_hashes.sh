#!/bin/bash
# Per-kernel VGPR count, spill counts and a hash of the emitted ISA of kernel sources (tools/isa_hashes.py does the work):
#   tools/isa_hashes.sh aero-optical-flow_amd/csrc/k_coarse.hip [more sources]
exec python3 "$(dirname "$0")/isa_hashes.py" "$@"
