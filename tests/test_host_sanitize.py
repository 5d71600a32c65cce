"""Host side of the storm.h API (containers, growth paths, marshalling towards the device, error
conventions) under AddressSanitizer + UBSan + LeakSanitizer on the CPU. The device is replaced by
tests/host_sanitize/device_stub.c (TEST ONLY: it returns the number of set bits that reached it,
not pair counts), so this needs no GPU; GPU sanitizers are not available on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_host_containers_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize"
    srcs = [os.path.join(ROOT, "stormbitmaps_amd", "csrc", "storm_host.c"),
            os.path.join(ROOT, "stormbitmaps_amd", "csrc", "storm_synth.c"),
                os.path.join(ROOT, "stormbitmaps_amd", "csrc", "storm_leaves.c"),
            os.path.join(ROOT, "tests", "host_sanitize", "device_stub.c"),
            os.path.join(ROOT, "tests", "host_sanitize", "driver.c")]
    build = subprocess.run(["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-Wall",
                            "-pthread", "-I" + os.path.join(ROOT, "include"), *srcs, "-o", str(exe), "-lm"],
                           capture_output=True, text=True)
    if build.returncode != 0 and "asan" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("libasan not installed")
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "host sanitize: ok" in run.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_host_side_with_three_caller_threads_under_tsan(tmp_path):
    """storm.h promises that different handles may be used from different threads: three threads, each with a
    STORM_t of 4500 rows (the arena fingerprint then runs on its helper threads) and a STORM_contiguous_t of its own,
    twenty all-pairs calls each, under ThreadSanitizer on the device stub — first behind ONE device slot, then with
    three slots configured and every thread narrowed to its own (STORM_hip_set_thread_devices: one lock per slot, the
    threads run side by side) while a fourth thread calls a raw-buffer wrapper (all slots locked)."""
    exe = tmp_path / "host_threads"
    srcs = [os.path.join(ROOT, "stormbitmaps_amd", "csrc", "storm_host.c"),
            os.path.join(ROOT, "stormbitmaps_amd", "csrc", "storm_synth.c"),
                os.path.join(ROOT, "stormbitmaps_amd", "csrc", "storm_leaves.c"),
            os.path.join(ROOT, "tests", "host_sanitize", "device_stub.c"),
            os.path.join(ROOT, "tests", "host_sanitize", "threads.c")]
    build = subprocess.run(["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=thread", "-pthread",
                            "-I" + os.path.join(ROOT, "include"), *srcs, "-o", str(exe), "-lm", "-ldl"],
                           capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("libtsan not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66"))
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "mt ok" in run.stdout and "ThreadSanitizer" not in run.stderr
