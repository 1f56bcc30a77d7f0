d=$(ls -d /sys/class/drm/card*/device | head -1)
echo DEV $d; ls $d/hwmon/hwmon*/ | tr '\n' ' '; echo
python tools/run_pairing.py pairing 16 60 > /dev/null 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  echo "t=$i sclk=$(grep '\*' $d/pp_dpm_sclk | tr -d '\n') freq1=$(cat $d/hwmon/hwmon*/freq1_input 2>/dev/null) power_avg=$(cat $d/hwmon/hwmon*/power1_average 2>/dev/null) power_in=$(cat $d/hwmon/hwmon*/power1_input 2>/dev/null) busy=$(cat $d/gpu_busy_percent)"
  sleep 0.2
done
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" 
wait $PID
