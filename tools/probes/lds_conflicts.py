"""LDS bank-conflict model of k_resblock_up's access patterns (lane groups of ds_read_b128 as in the CDNA guide; writes: 128 bytes per cycle in lane order)
and a search over swizzles / row pitches: python tools/probes/lds_conflicts.py.  Units: LDS cycles per wave-instruction (ideal: 4; 8 for a b128 write)."""
# bank-conflict model (cdna guide): ds_read_b128: 64 banks, lane groups below; ds_write_b64: 32 banks, halves of 32 lanes? -> model 16 lanes/cycle in order;
# ds_write_b128: 32 banks, 8 lanes/cycle in order.  cost = sum over groups of max bank multiplicity (in units of cycles per group)
import itertools
G128=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128=G128+[[l+32 for l in g] for g in G128]
def cost_read128(addr):  # addr(lane)->byte
    tot=0
    for g in G128:
        cnt={}
        for l in g:
            b0=(addr(l)//4)%64
            for k in range(4): cnt[(b0+k)%64]=cnt.get((b0+k)%64,0)+1
        tot+=max(cnt.values())
    return tot  # ideal 4
def cost_write(addr,nbytes):
    per=128//nbytes  # lanes per cycle
    tot=0
    for s in range(0,64,per):
        cnt={}
        for l in range(s,s+per):
            b0=(addr(l)//4)%32
            for k in range(nbytes//4): cnt[(b0+k)%32]=cnt.get((b0+k)%32,0)+1
        tot+=max(cnt.values())
    return tot  # ideal 64/per
esw=lambda r: r^(r>>2)
usw=lambda i:(i^(i>>2))&15
def report(name,costs,ideal): print(f"{name:38s} avg {sum(costs)/len(costs):6.2f} (ideal {ideal}) max {max(costs)}")
# stage 0 x-plane reads
c=[]
for s in range(8):
  for rt in range(2):
    jr=s>>2; c0=(s&3)*32
    def addr(l,rt=rt,jr=jr,c0=c0):
        r16=l&15; g=l>>4; rho=rt*16+r16+jr
        return rho*256+((((c0>>3)+g)^rho)&15)*16
    c.append(cost_read128(addr))
report("stage0 x reads (k^rho)",c,4)
c=[]
for s in range(8):
  for rt in range(2):
    jr=s>>2; c0=(s&3)*32
    def addr(l,rt=rt,jr=jr,c0=c0):
        r16=l&15; g=l>>4; rho=rt*16+r16+jr
        return rho*256+((((c0>>3)+g)^esw(rho))&15)*16
    c.append(cost_read128(addr))
report("stage0 x reads (k^esw(rho))",c,4)
# stage B eu reads, C=64 rows 128 B, CM=7
for f,name in ((lambda r:r,"rho"),(esw,"esw")):
  c=[]
  for wave in range(8):
    for s in range(6):
      tap=(s*32)//64; c0=(s*32)%64
      def addr(l,wave=wave,tap=tap,c0=c0,f=f):
          r16=l&15; g=l>>4; rho=wave*16+r16+tap
          return rho*128+((((c0>>3)+g)^f(rho))&7)*16
      c.append(cost_read128(addr))
  report(f"stage B/D eu reads ({name})",c,4)
# stage 0 eu writes (b64): lane rows i=4*(rt*16+r16)+g, ch=8w+4ct
for f,name in ((lambda r:r,"rho"),(esw,"esw")):
  c=[]
  for wave in range(8):
    for rt in range(2):
      for ct in range(2):
        ch=8*wave+4*ct
        def addr(l,rt=rt,ch=ch,f=f):
            r16=l&15; g=l>>4; i=4*(rt*16+r16)+g; rho=i+2
            return rho*128+((((ch>>3)^f(rho))&7)<<4)+((ch&4)<<1)
        c.append(cost_write(addr,8))
  report(f"stage 0 eu writes b64 ({name})",c,4)
# stage C eu writes b64: lane row i_lane=wave*16+r16, ch=n*16+4g
for f,name in ((lambda r:r,"rho"),(esw,"esw")):
  c=[]
  for wave in range(8):
    for n in range(4):
      def addr(l,wave=wave,n=n,f=f):
          r16=l&15; g=l>>4; rho=wave*16+r16+2; ch=n*16+4*g
          return rho*128+((((ch>>3)^f(rho))&7)<<4)+((ch&4)<<1)
      c.append(cost_write(addr,8))
  report(f"stage C eu writes b64 ({name})",c,4)
# ut writes b128 (stage 0) and reads (stage C)
c=[]
for wave in range(8):
  for rt in range(2):
    for ct in range(2):
      ch=8*wave+4*ct
      def addr(l,rt=rt,ch=ch):
          r16=l&15; g=l>>4; i=4*(rt*16+r16)+g
          return i*256+(((ch>>2)^usw(i))<<4)
      c.append(cost_write(addr,16))
report("ut writes b128",c,8)
c=[]
for wave in range(8):
  for n in range(4):
    def addr(l,wave=wave,n=n):
        r16=l&15; g=l>>4; i=wave*16+r16; ch=n*16+4*g
        return i*256+(((ch>>2)^usw(i))<<4)
    c.append(cost_read128(addr))
report("ut reads b128",c,4)
# stage X writes b64: e=tid+j*512 -> r=e//32,c=(e%32)*4
c=[]
for w in range(8):
  for j in range(2):
    def addr(l,w=w,j=j):
        e=w*64+l+j*512; r=e//32; cc=(e%32)*4
        return r*256+((((cc>>3)^r)&15)<<4)+((cc&4)<<1)
    c.append(cost_write(addr,8))
report("stage X writes b64",c,4)
# stage B h writes b64 and stage C h reads
c=[]
for wave in range(8):
  for n in range(2):
    def addr(l,wave=wave,n=n):
        r16=l&15; g=l>>4; i=wave*16+r16; ch=n*16+4*g
        return i*64+((((ch>>3)^i)&3)<<4)+((ch&4)<<1)
    c.append(cost_write(addr,8))
report("stage B h writes b64",c,4)
c=[]
for wave in range(8):
    def addr(l,wave=wave):
        r16=l&15; g=l>>4; i=wave*16+r16
        return i*64+(((0*4+g)^i)&3)*16
    c.append(cost_read128(addr))
report("stage C h reads b128",c,4)
# weight fragment reads from LDS: contiguous 1 KB per wave
c=[cost_read128(lambda l:l*16)]
report("weight frag reads (contiguous)",c,4)
print("---- search")
def fam():
    for s1 in (0,1,2,3,4):
        for s2 in (0,1,2,3,4,5):
            for m in (1,3,5,7):
                yield (s1,s2,m), (lambda r,s1=s1,s2=s2,m=m: ((r*m) ^ ((r>>s1) if s1 else 0) ^ ((r>>s2) if s2 else 0)))
def eu_cost(f):
    tot=0
    for wave in range(8):
        for s in range(6):
            tap=(s*32)//64; c0=(s*32)%64
            tot+=2*cost_read128(lambda l: (lambda r16,g,rho: rho*128+((((c0>>3)+g)^f(rho))&7)*16)(l&15,l>>4,wave*16+(l&15)+tap))   # B and D
        for rt in range(2):
            for ct in range(2):
                ch=8*wave+4*ct
                tot+=cost_write(lambda l: (lambda rho: rho*128+((((ch>>3)^f(rho))&7)<<4)+((ch&4)<<1))(4*(rt*16+(l&15))+(l>>4)+2),8)*2  # hi+lo
        for n in range(4):
            tot+=cost_write(lambda l: (lambda rho,ch: rho*128+((((ch>>3)^f(rho))&7)<<4)+((ch&4)<<1))(wave*16+(l&15)+2,n*16+4*(l>>4)),8)*2
    return tot/8
best=sorted((eu_cost(f),k) for k,f in fam())[:6]
print("eu planes (per wave): rho", eu_cost(lambda r:r), "esw", eu_cost(esw), "best", best)
def x_cost(f):
    tot=0
    for s in range(8):
        for rt in range(2):
            jr=s>>2; c0=(s&3)*32
            tot+=2*cost_read128(lambda l: (lambda rho,g: rho*256+((((c0>>3)+g)^f(rho))&15)*16)(rt*16+(l&15)+jr,l>>4))
    w=0
    for ww in range(8):
        for j in range(2):
            w+=2*cost_write(lambda l: (lambda r,cc: r*256+((((cc>>3)^f(r))&15)<<4)+((cc&4)<<1))((ww*64+l+j*512)//32,((ww*64+l+j*512)%32)*4),8)
    return tot+w/8
best=sorted((x_cost(f),k) for k,f in fam())[:6]
print("x planes (per wave): r", x_cost(lambda r:r), "best", best)
def h_cost(f):
    tot=0
    for wave in range(8):
        for n in range(2):
            tot+=2*cost_write(lambda l: (lambda i,ch: i*64+((((ch>>3)^f(i))&3)<<4)+((ch&4)<<1))(wave*16+(l&15),n*16+4*(l>>4)),8)
        tot+=2*cost_read128(lambda l: (lambda i,g: i*64+(((g)^f(i))&3)*16)(wave*16+(l&15),l>>4))
    return tot/8
best=sorted((h_cost(f),k) for k,f in fam())[:6]
print("h planes (per wave): i", h_cost(lambda r:r), "best", best)
print("---- padded pitches (no XOR)")
def eu_cost_p(P):
    tot=0
    for wave in range(8):
        for s in range(6):
            tap=(s*32)//64; c0=(s*32)%64
            tot+=2*cost_read128(lambda l: (wave*16+(l&15)+tap)*P+((c0>>3)+(l>>4))*16)
        for rt in range(2):
            for ct in range(2):
                ch=8*wave+4*ct
                tot+=cost_write(lambda l: (4*(rt*16+(l&15))+(l>>4)+2)*P+ch*2,8)*2
        for n in range(4):
            tot+=cost_write(lambda l: (wave*16+(l&15)+2)*P+(n*16+4*(l>>4))*2,8)*2
    return tot/8
print("eu:", [(P,eu_cost_p(P)) for P in (128,136,144,160,176,192,208,224,272,288)])
def x_cost_p(P):
    tot=0
    for s in range(8):
        for rt in range(2):
            jr=s>>2; c0=(s&3)*32
            tot+=2*cost_read128(lambda l: (rt*16+(l&15)+jr)*P+((c0>>3)+(l>>4))*16)
    w=0
    for ww in range(8):
        for j in range(2):
            w+=2*cost_write(lambda l: ((ww*64+l+j*512)//32)*P+((ww*64+l+j*512)%32)*8,8)
    return tot+w/8
print("x:", [(P,x_cost_p(P)) for P in (256,272,288,304,320)])
def h_cost_p(P):
    tot=0
    for wave in range(8):
        for n in range(2):
            tot+=2*cost_write(lambda l: (wave*16+(l&15))*P+(n*16+4*(l>>4))*2,8)
        tot+=2*cost_read128(lambda l: (wave*16+(l&15))*P+(l>>4)*16)
    return tot/8
print("h:", [(P,h_cost_p(P)) for P in (64,72,80,96,112,144)])
