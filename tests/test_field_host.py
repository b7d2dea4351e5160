"""Host side of the library's field arithmetic, without a GPU: the binary extended-GCD inversion (spartan2_amd/csrc/field.hpp fe_inv_host_xgcd, raw for public values and behind a multiplicative mask in fe_inv — every
normalisation of a point on the host and the prover's division by 1 - r_y[0] go through it) against Fermat's little theorem and x * inv(x) == 1, on
edge values and seeded random residues of both fields (tests/native/inv_check.hip, compiled here with hipcc: host code only, nothing is launched)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_host_inversion_matches_fermat(tmp_path):
    exe = str(tmp_path / "inv_check")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-o", exe, os.path.join(HERE, "native", "inv_check.hip")], check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "scalar field: 20000 samples, 0 mismatches" in out.stdout and "base field: 20000 samples, 0 mismatches" in out.stdout


def _build_mul_check(tmp_path):
    exe = str(tmp_path / "mul_check")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-w", "-o", exe, os.path.join(HERE, "native", "mul_check.hip")], check=True, capture_output=True, timeout=600)
    return exe


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_product_scanning_base_field_product_on_the_host(tmp_path):
    """The base-field product the kernels use (column-wise 96-bit accumulation, field.hpp / field_fips_device.hpp) in its plain-C form and the 32-bit
    row-wise form against the 64-bit CIOS product: edge values and seeded random residues."""
    out = subprocess.run([_build_mul_check(tmp_path), "host", "50000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "host: 0 mismatches in 50000 products" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_product_scanning_base_field_product_on_the_device(tmp_path):
    """The generated asm column blocks on the device (a lone wave, then full blocks; single products and 64-deep dependent chains) against the host."""
    out = subprocess.run([_build_mul_check(tmp_path), "device", "100000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "device: 0 mismatches in 100064 lanes" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_host_keccak_permutation_forms_agree(tmp_path):
    """The unrolled host permutation (generic and BMI builds) against the loop form the device compiles, plus the Keccak-256 digests of "" and "abc"."""
    exe = str(tmp_path / "keccak_check")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-w", "-o", exe, os.path.join(HERE, "native", "keccak_check.hip")], check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "keccak: 0 mismatches" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_limb_sum_reduction_matches_modular_additions(tmp_path):
    """fe_from_limb_sums (the one reduction behind the host's limb-wise gathering of up to 64 result slots per round) against a chain of modular additions:
    1..64 random, all-ones, zero and near-p summands, both fields."""
    exe = str(tmp_path / "limb_sum_check")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-w", "-o", exe, os.path.join(HERE, "native", "limb_sum_check.hip")], check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "scalar field: 20000 sums, 0 mismatches" in out.stdout and "base field: 20000 sums, 0 mismatches" in out.stdout
