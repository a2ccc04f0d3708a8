"""Sample the shader clock / power while a training loop runs: python tools/sclk_sample.py  (prints min / median / max MHz)"""
import os, sys, subprocess, time, re, glob, statistics
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
p = subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--steps', '800', '--warmup', '10'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
clk, pw = [], []
t0 = time.time()
while p.poll() is None and time.time() - t0 < 120:
    try:
        out = subprocess.run(['/opt/rocm/bin/rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r'sclk clock level: \w+: \((\d+)Mhz\)', out)
        if m: clk.append(int(m.group(1)))
        m = re.search(r'Power \(W\): ([\d.]+)', out)
        if m: pw.append(float(m.group(1)))
    except Exception as e:
        print('smi failed', e); break
    time.sleep(0.05)
out = p.communicate()[0].decode()
print('samples', len(clk), 'sclk MHz min/median/max', min(clk) if clk else None, statistics.median(clk) if clk else None, max(clk) if clk else None)
print('power W min/median/max', min(pw) if pw else None, statistics.median(pw) if pw else None, max(pw) if pw else None)
print(clk)
print(pw)
import json
try:
    d = json.loads(out.strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])
except Exception as e:
    print('bench output', out[-300:])
