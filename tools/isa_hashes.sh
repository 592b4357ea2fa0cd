#!/bin/bash
# Per-kernel VGPR count, spill counts and a hash of the emitted ISA of one kernel source:
#   tools/isa_hashes.sh aero-optical-flow_amd/csrc/k_search_lane8.hip [extra hipcc flags]
# Two builds whose hash of a kernel agrees run the same instructions for it.  (The code hipcc emits for a
# kernel depends on what else the translation unit inlines -- LAB_LOG.md, round 4 -- so a change that
# "cannot" touch the headline kernel is checked here before it goes to the GPU.)
set -e
src=$1; shift
tmp=$(mktemp -d)
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -I$root/include -I$root/aero-optical-flow_amd/csrc --offload-arch=gfx950 -fno-fast-math \
    -S --offload-device-only "$@" $src -o $tmp/out.s 2>/dev/null
python3 - $tmp/out.s <<'PY'
import hashlib, re, subprocess, sys
s = open(sys.argv[1]).read()
meta = {}
for blk in re.findall(r'- \.agpr_count.*?\.wavefront_size:\s+\d+', s, re.S):
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    meta[g('name')] = (g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'))
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    body = '\n'.join(l for l in m.group(2).split('\n') if not l.strip().startswith(';'))
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'aof::\(anonymous namespace\)::', '', name)
    name = re.sub(r'\(.*', '', name)
    v, vs, ss, scratch = meta.get(m.group(1), ('?',) * 4)
    print(f'{hashlib.md5(body.encode()).hexdigest()[:10]}  vgpr {v:>3s}  vgpr spills {vs}  sgpr spills {ss}  scratch {scratch:>3s} B  {name}')
PY
rm -rf $tmp
