import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import build
import subprocess, re, sys
f=sys.argv[1]; pat=sys.argv[2]
cmd=["hipcc","--offload-arch=gfx950","-c","faststyle_amd/csrc/"+f,"-o","/tmp/g.o","-Rpass-analysis=kernel-resource-usage"]+build.FLAGS+build.FILE_FLAGS.get(f,[])+["-Iinclude","-Ifaststyle_amd/csrc"]+sys.argv[3:]
r=subprocess.run(cmd,capture_output=True,text=True)
name=None
for l in r.stderr.splitlines():
    m=re.search(r"Function Name: (\S+)",l)
    if m: name=m.group(1)
    m=re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]): (\d+)",l)
    if m and name and pat in name: print(name[:70],m.group(1),m.group(2))
print(r.returncode, r.stderr[-600:] if r.returncode else "")
