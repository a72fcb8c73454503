"""Puts a measurement set's tables into DESIGN.md (between the R6_* markers) and the headline numbers into README.md:
python scratch/fill_design.py <tag>"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, 'gpurun_out', tag)
table = subprocess.run([sys.executable, os.path.join(ROOT, 'scratch', 'r6_table.py'), tag], capture_output=True, text=True).stdout
host = open(os.path.join(out, 'host_budget_two_cpus.md')).read()
path = os.path.join(ROOT, 'DESIGN.md')
text = open(path).read()
text = re.sub(r'<!-- R6_TABLE_BEGIN -->.*?<!-- R6_TABLE_END -->',
              lambda m: f'<!-- R6_TABLE_BEGIN -->\nSet `{tag}` (`scratch/measure_r6.sh {tag}`, one gpurun call):\n\n' + table + '<!-- R6_TABLE_END -->', text, flags=re.S)
text = re.sub(r'<!-- R6_HOST_BEGIN -->.*?<!-- R6_HOST_END -->',
              lambda m: '<!-- R6_HOST_BEGIN -->\n' + '\n'.join('  ' + line for line in host.strip().splitlines()) + '\n<!-- R6_HOST_END -->', text, flags=re.S)
open(path, 'w').write(text)


def line(name):
    return json.load(open(os.path.join(out, name + '.json')))


head, small, age, driving = line('bench'), line('bench_224x224'), line('bench_age_vgg64_bf16'), line('bench_driving_64x192_fp16')
values = {'R6_HEADLINE': f"{head['value']:.1f}", 'R6_FRAC': f"{100 * head['roofline']['frac']:.1f} %", 'R6_224': f"{small['value']:.1f}",
          'R6_AGE_FRAC': f"{age['roofline']['frac']:.3f}", 'R6_DRIVING_FRAC': f"{driving['roofline']['frac']:.3f}",
          'R6_AGE': f"{age['value']:.0f}", 'R6_DRIVING': f"{driving['value']:.0f}"}
print(values)
path = os.path.join(ROOT, 'README.md')
text = open(os.path.join(ROOT, 'scratch', 'README.template.md')).read()          # (README.md with R6_* placeholders)
for key in sorted(values, key=len, reverse=True):
    text = text.replace(key, values[key])
open(path, 'w').write(text)
