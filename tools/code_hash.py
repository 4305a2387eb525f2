"""Hash of every kernel of a built library (disassembly of its gfx950 code objects, addresses stripped): `python tools/code_hash.py lib.so`.
What a claim like "the built-in kernels are unchanged byte for byte by this feature" is checked with (diff two outputs)."""
import struct, sys, hashlib, subprocess, tempfile, re
def blobs(path):
    data=open(path,'rb').read(); out=[]; start=0
    while True:
        i=data.find(b"__CLANG_OFFLOAD_BUNDLE__", start)
        if i<0: break
        n,=struct.unpack_from("<Q",data,i+24); pos=i+32; end=pos
        for _ in range(n):
            off,size,idlen=struct.unpack_from("<QQQ",data,pos); pos+=24
            tid=data[pos:pos+idlen].decode(); pos+=idlen
            if "gfx950" in tid: out.append(data[i+off:i+off+size])
            end=max(end,i+off+size)
        start=max(end,i+24)
    return out
res={}
for b in blobs(sys.argv[1]):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(b); f.flush()
        txt=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump","-d","--no-show-raw-insn",f.name],capture_output=True,text=True).stdout
    cur=None
    for line in txt.splitlines():
        m=re.match(r"^[0-9a-f]+ <(.+)>:$",line)
        if m: cur=m.group(1); res[cur]=hashlib.sha256(); continue
        if cur and line.strip():
            # strip addresses
            res[cur].update(re.sub(r"^\s*[0-9a-f]+:\s*","",line).encode())
for k in sorted(res): print(k, res[k].hexdigest()[:16])
