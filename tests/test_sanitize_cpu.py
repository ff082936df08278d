"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the product (review item 6 of round 2): tcv_pack.cpp,
tcv_estimator.cpp and the host halves of tcv_capi.hip / tcv_marg.hip / the other .hip files, built by
`python tc-viml_amd/build.py --sanitize` (hipcc: the host pass is instrumented, the gfx950 pass is not -- GPU sanitizers are not
available on this pool).  The instrumented library is driven without a device, in a child interpreter with the ASan runtime
preloaded: the packer through tcv_problem_plan_stats / tcv_problem_from_window / tcv_prior_create on the golden windows, the
800-block window, the ragged windows and every malformed input of tests/test_abi_cpu.py, and the native estimator's window
management up to the point where it asks for the device."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_mod():
    spec = importlib.util.spec_from_file_location("tcv_build", os.path.join(ROOT, "tc-viml_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_host_side_is_clean_under_asan_and_ubsan(built):
    mod = _build_mod()
    rt = mod.asan_runtime()
    assert rt, "ROCm clang's shared ASan runtime not found"
    lib = mod.build(sanitize=True)
    env = dict(os.environ, TCV_LIB=lib, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=97",      # CPython itself leaks by design
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_pack_cpu.py"), os.path.join(ROOT, "tests", "test_abi_cpu.py"),
           os.path.join(ROOT, "tests", "test_replay_cpu.py") + "::test_native_estimator_fails_loudly_without_a_device",
           # links an example against -ltcv_hip by file name and runs it outside the preload: covered by the plain suite
           "--deselect", os.path.join(ROOT, "tests", "test_abi_cpu.py") + "::test_header_is_valid_c99_and_cxx_and_the_example_links"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-4000:]
    assert "runtime error" not in text and "AddressSanitizer" not in text, text[-4000:]
    assert " passed" in text and "failed" not in text
    # the child really ran against the instrumented library
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, 'tc-viml_amd'); import tcv; tcv.lib(); print(open('/proc/self/maps').read().count('libtcv_hip_san.so') > 0)"],
                           capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert probe.stdout.strip().endswith("True"), probe.stdout + probe.stderr
