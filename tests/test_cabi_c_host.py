"""The C ABI from C: include/qatiger.h must compile as plain C, and a C host program (tests/c/cabi_host.c, no torch, no
C++) must be able to drive the kernels with hipMalloc'ed buffers."""
import os
import subprocess

import pytest

from conftest import REPO

INC = os.path.join(REPO, 'include')
LIBDIR = os.path.join(REPO, 'qa-tiger_amd')


def test_header_is_valid_c(tmp_path):
    src = tmp_path / 'h.c'
    src.write_text('#include "qatiger.h"\nint main(void) { int (*f)(void) = qt_version; (void)f; return 0; }\n')
    res = subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-fsyntax-only', f'-I{INC}', str(src)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


@pytest.mark.gpu
def test_c_host_program_runs_the_kernels(tmp_path, qt):
    qt._lib.load()                                               # makes sure the library is built
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    exe = str(tmp_path / 'cabi_host')
    build = subprocess.run(['gcc', '-std=c99', '-O1', os.path.join(REPO, 'tests', 'c', 'cabi_host.c'), '-D__HIP_PLATFORM_AMD__',
                            f'-I{rocm}/include', f'-I{INC}', f'-L{LIBDIR}', '-lqatiger_hip', f'-L{rocm}/lib', '-lamdhip64',
                            f'-Wl,-rpath,{LIBDIR}', f'-Wl,-rpath,{rocm}/lib', '-lm', '-o', exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and 'OK' in run.stdout, run.stdout + run.stderr
