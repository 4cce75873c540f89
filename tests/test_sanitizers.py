"""AddressSanitizer + UndefinedBehaviorSanitizer over the native HOST code (GPU sanitizers are not available on the pool, so this
is the CPU build only): every piece of C / C++ that a host-side test touches is rebuilt with `-fsanitize=address,undefined` and the tests that drive
it are re-run in a child process under LD_PRELOAD=libasan --
  * booster_gym_amd/csrc/bg_urdf.cpp + bg_model.cpp (the asset loader of the C ABI): the synthetic T1 URDF, the reference's URDF where the tree is
    present, and the malformed files (bad axis, truncated XML, missing file, a link with two parents, a joint cycle, a self-joint, XML nested 500 deep,
    a foot box of the wrong shape);
  * tests/host_harness/{harness,rng_harness}.cpp: the product's per-lane dynamics headers (one leg per lane and the packed one-env-per-lane form) and its
    Philox header compiled for the host, 150 states per case against the float64 oracle, body contacts, height field and crossed legs included;
  * oracle/dyn_ref.c: the oracle itself, in the comparisons above and stepped through contact (standing, crossed legs, a whole walking episode).
A sanitizer report aborts the child (halt_on_error, -fno-sanitize-recover) and fails this test with the report in the message."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_native_host_code_is_clean_under_asan_and_ubsan():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()

    def is_elf(path):  # (on RHEL-like distributions libasan.so is a linker script that LD_PRELOAD cannot load; some hosts have none at all)
        try:
            with open(path, "rb") as f:
                return f.read(4) == b"\x7fELF"
        except OSError:
            return False

    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    if not (os.path.isabs(asan) and is_elf(os.path.realpath(asan)) and os.path.isabs(ubsan) and is_elf(os.path.realpath(ubsan))):
        pytest.skip("no loadable libasan.so / libubsan.so next to gcc on this host")
    # libstdc++ beside it: the runtime resolves __cxa_throw when it starts, and python itself does not link the C++ library (the loader throws inside)
    cxx = subprocess.check_output(["gcc", "-print-file-name=libstdc++.so.6"], text=True).strip()
    env = dict(os.environ, BG_SANITIZE="1", LD_PRELOAD=asan + " " + cxx, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2", PYTHONPATH=ROOT)
    select = ("test_product_rng_header_matches_oracle_philox or test_product_dynamics_header_matches_oracle or test_c_urdf_loader_matches_the_python_loader "
              "or test_asset_without_leg_collision_shapes_loads_with_self_collision_off or test_c_urdf_loader_on_the_reference_asset "
              "or test_self_collision_pushes_crossed_feet_apart or test_standing_normal_force or test_trained_reference_policy_walks_in_the_oracle")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_host_logic.py"), os.path.join(HERE, "test_oracle_golden.py"), "-x", "-q",
                        "-p", "no:cacheprovider", "-k", select], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = p.stdout[-3000:] + p.stderr[-6000:]
    assert p.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # (every selected test ran: none of them was silently deselected by a rename)
    last = [l for l in p.stdout.splitlines() if " passed" in l][-1]
    # exactly the eight selected tests where the reference tree is present (test_c_urdf_loader_on_the_reference_asset skips without it): a rename that
    # silently deselects one of them fails here
    want = 8 if os.path.isdir("/root/reference") else 7
    assert int(last.split(" passed")[0].split()[-1]) == want, (want, last)
