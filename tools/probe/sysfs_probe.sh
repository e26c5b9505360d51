for c in /sys/class/drm/card*/device; do echo "== $c"; ls $c | tr '\n' ' '; echo; for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk gpu_busy_percent mem_busy_percent current_link_speed; do echo "-- $f"; cat $c/$f 2>&1 | head -12; done; for h in $c/hwmon/hwmon*; do echo "== $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap temp1_input temp2_input temp3_input freq1_input freq2_input; do echo "-- $f: $(cat $h/$f 2>&1)"; done; done; done
python3 -c "
import time
t=time.time()
try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    print('amdsmi ok', time.time()-t)
    for fn, args in (('amdsmi_get_clock_info', (h, amdsmi.AmdSmiClkType.GFX)), ('amdsmi_get_clock_info', (h, amdsmi.AmdSmiClkType.MEM)), ('amdsmi_get_power_info', (h,)), ('amdsmi_get_gpu_metrics_info', (h,)), ('amdsmi_get_power_cap_info', (h,))):
        t=time.time()
        try:
            print(fn, getattr(amdsmi, fn)(*args), 'took', time.time()-t)
        except Exception as e:
            print(fn, 'ERR', e)
except Exception as e:
    print('amdsmi unavailable', e)
"
