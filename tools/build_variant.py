#!/usr/bin/env python3
"""A/B builds: compile ONE source of the library with an extra -D and link it with the in-tree objects of the rest.
usage: python tools/build_variant.py <tag> <source.hip> <DEFINE[=value]|none>   ->  ddsp_piano_amd/libddspp_<tag>.so
(select it with DDSPP_LIB=<path>; the in-tree build must be current: python -c "import __graft_entry__ as g; g.build()")"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddsp_piano_amd import _lib  # noqa: E402

tag, source, define = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join(_lib._CSRC, source)
stem = os.path.splitext(source)[0]
obj = f'/tmp/{stem}_{tag}.o'
flags = [f'--offload-arch={_lib.ARCH}', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-I', _lib._CSRC,
         '-I', os.path.join(os.path.dirname(_lib._HERE), 'include'), '-w'] + _lib.PER_FILE_FLAGS.get(source, [])
subprocess.run([_lib._hipcc()] + flags + ([f'-D{define}'] if define != 'none' else []) + ['-x', 'hip', '-c', src, '-o', obj], check=True)
objs = [os.path.join(_lib._HERE, 'build', os.path.splitext(s)[0] + '.o') for s in _lib.SOURCES]
objs = [obj if os.path.basename(o) == stem + '.o' else o for o in objs]
out = os.path.join(_lib._HERE, f'libddspp_{tag}.so')
subprocess.run([_lib._hipcc(), f'--offload-arch={_lib.ARCH}', '-shared', '-fPIC'] + objs +
               ['-L', '/opt/rocm/lib', '-lrocfft', '-Wl,-rpath,/opt/rocm/lib', '-o', out], check=True)
print(out)
