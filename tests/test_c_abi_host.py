"""GPU: the boundary is a C ABI — drive it from a plain C program (no Python, no torch in the process)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_c_host_drives_the_library(tmp_path):
    exe = str(tmp_path / 'c_abi_host')
    lib, orc = os.path.join(ROOT, 'ursabench_amd', 'csrc'), os.path.join(ROOT, 'oracle')
    subprocess.run(['gcc', '-O2', '-std=c11', '-D__HIP_PLATFORM_AMD__', os.path.join(ROOT, 'tests', 'c_abi_host.c'),
                    '-I' + os.path.join(ROOT, 'include'), '-I/opt/rocm/include', '-L' + lib, '-lursa_hip', '-L' + orc,
                    '-loracle', '-L/opt/rocm/lib', '-lamdhip64', '-lm', f'-Wl,-rpath,{lib}', f'-Wl,-rpath,{orc}',
                    '-Wl,-rpath,/opt/rocm/lib', '-o', exe], check=True, capture_output=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'C-ABI host OK' in out.stdout, out.stdout + out.stderr
