import ctypes as C, glob, os, sys, time
sys.path.insert(0, os.getcwd())
import starkpack_winterfell_amd.capi as capi
from starkpack_winterfell_amd.build import yardstick_path
capi.load()
Y = C.CDLL(yardstick_path())
out = (C.c_double * 9)()
print("yardstick rc", Y.wf_yardstick_run(0, out), list(out))
for pat in ("/sys/class/drm/card*/device/pp_dpm_sclk", "/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "/sys/class/drm/card*/device/hwmon/hwmon*/freq*_label", "/sys/class/drm/card*/device/gpu_busy_percent"):
    for f in glob.glob(pat):
        try:
            print(f, "->", open(f).read().strip().replace("\n", " | "))
        except Exception as e:
            print(f, "unreadable:", e)
