"""libsquidstitch from plain C (no Python objects, no torch in the process): compile the C program
with gcc against include/squidstitch.h, link the shared library, run it on the GPU."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_links_and_runs(tmp_path):
    gcc = shutil.which('gcc') or 'gcc'
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    libdir = os.path.join(ROOT, 'image-stitcher_amd', 'csrc')
    exe = str(tmp_path / 'c_abi_smoke')
    subprocess.run([gcc, '-std=c11', '-D__HIP_PLATFORM_AMD__', os.path.join(ROOT, 'tests', 'c_abi', 'c_abi_smoke.c'),
                    '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(rocm, 'include'),
                    '-L', libdir, '-lsquidstitch', '-L', os.path.join(rocm, 'lib'), '-lamdhip64',
                    '-Wl,-rpath,' + libdir, '-Wl,-rpath,' + os.path.join(rocm, 'lib'), '-o', exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'c-abi smoke ok' in out.stdout
