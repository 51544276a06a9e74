import json,sys,subprocess,os
for stop in (0,1,2,3):
    env=dict(os.environ, BTGPU_WIN_STOP=str(stop))
    out=subprocess.run([sys.executable,'bench.py','--no-cpu'],env=env,capture_output=True,text=True).stdout.strip().split('\n')[-1]
    try:
        r=json.loads(out); print(stop, r['roofline']['kernel_avg_ms']['window'], r['ms_per_step'])
    except Exception as e: print(stop, 'ERR', out[-300:])
