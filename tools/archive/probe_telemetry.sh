set -x
mkdir -p gpurun_out/r05_base
id
ls -la /sys/class/drm/ 2>&1 | head -30
for c in /sys/class/drm/card*/device; do echo "== $c"; ls $c | tr '\n' ' '; echo; ls $c/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; done 2>&1 | head -60
for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/hwmon/*/power1_average /sys/class/drm/card*/device/hwmon/*/power1_input /sys/class/drm/card*/device/hwmon/*/power1_cap /sys/class/drm/card*/device/hwmon/*/freq1_input; do echo "-- $f"; cat $f 2>&1 | head -12; done
timeout 60 amd-smi metric --clock --power --json 2>&1 | head -120
timeout 60 amd-smi static --limit 2>&1 | head -60
timeout 60 rocm-smi --showclocks --showpower 2>&1 | head -60
