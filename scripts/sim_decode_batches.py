"""Offline: how the decoder's batches look on the bench's data -- sequences per 64-byte input window, how many have their source
inside the batch, how many dependency rounds the copy rule needs and how deep the true dependency chains are, for the present
batches (one window) and for batches of several windows (bigger batches do not amortise the rounds).  CPU only;
the sequences are parsed from the reference encoder's output (oracle/_ref)."""
import sys; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np
from plz4_amd import synth
from orclib import Ref, Oracle
ref=Ref(); orc=Oracle()
src=synth.text(4<<20)
n,comp=ref.compress_fast(src, orc.bound(src.size)) if hasattr(ref,'compress_fast') else None
comp=np.asarray(comp[:n]) if n!=len(comp) else comp
c=bytes(comp)
# parse sequences
seqs=[]  # (ip_tok, ll, ml, off, nxt_ip, plain)
ip=0; N=len(c)
while ip<N:
    t=c[ip]; p=ip; ip+=1
    ll=t>>4
    if ll==15:
        while True:
            b=c[ip]; ip+=1; ll+=b
            if b!=255: break
    litp=ip; ip+=ll
    if ip>=N: seqs.append((p,ll,0,0,ip)); break
    off=c[ip]|(c[ip+1]<<8); ip+=2
    ml=t&15
    if ml==15:
        while True:
            b=c[ip]; ip+=1; ml+=b
            if b!=255: break
    ml+=4
    seqs.append((p,ll,ml,off,ip))
print("sequences",len(seqs),"comp",N, "avg in/seq",N/len(seqs),"avg out/seq",src.size/len(seqs))
S=np.array(seqs,dtype=np.int64)
ll=S[:,1]; ml=S[:,2]; off=S[:,3]
print("ll>=14:",(ll>=14).mean(),"ml>=19:",(ml>=19).mean(), "ml>273", (ml>273).mean(), "off<ml (overlap)",(off<ml).mean())
outLen=ll+ml; outStart=np.cumsum(outLen)-outLen
def plain(i): return ll[i]<14 and ml[i]<=273
def rounds(lo,hi):
    """near rounds for sequences [lo,hi) batch whose output starts at outStart[lo]; returns (#far, #near_rounds, #coop)"""
    op0=outStart[lo]
    idx=np.arange(lo,hi)
    mst=outStart[idx]+ll[idx]      # match dst start
    sp=mst-off[idx]
    coop=(ml[idx]>18)|(off[idx]<ml[idx])
    far=(~coop)&(sp+ml[idx]<=op0)
    pend=list(np.nonzero(~far)[0])
    r=0;nco=0
    while pend:
        f=pend[0]
        if coop[f]:
            nco+=1; pend.pop(0); continue
        lo_=mst[f]
        go=[j for j in pend if (not coop[j]) and sp[j]+ml[idx][j]<=lo_]
        r+=1
        gs=set(go); pend=[j for j in pend if j not in gs]
    return int(far.sum()), r, nco
# current scheme: window of 64 input bytes
def sim(mode,cap=64,maxout=1024,maxwin=6):
    i=0;nb=0;tr=0;tco=0;tw=0;tm=0;seqstep=0
    while i<len(seqs)-1:
        if not plain(i): seqstep+=1;i+=1;continue
        j=i;ip0=S[i,0];w=0
        if mode==0:
            while j<len(seqs)-1 and plain(j) and S[j,0]-ip0<64 and (S[j,0]+1+ll[j]-ip0)<=64 and outStart[j]+outLen[j]-outStart[i]<=1024: j+=1
            w=1
        else:
            wstart=ip0;w=1
            while j<len(seqs)-1 and plain(j) and (j-i)<cap and outStart[j]+outLen[j]-outStart[i]<=maxout:
                if S[j,0]-wstart>=64:
                    if w==maxwin: break
                    wstart=S[j,0]; w+=1
                j+=1
        if j==i: seqstep+=1;i+=1;continue
        f,r,nco=rounds(i,j)
        nb+=1;tr+=r;tco+=nco;tw+=w;tm+=j-i
        i=j
    print(f"mode {mode} cap {cap}: batches {nb} windows {tw} members/batch {tm/nb:.1f} near rounds/batch {tr/nb:.2f} coop/batch {tco/nb:.2f} seqsteps {seqstep}  | per 64B-window: rounds {tr/tw:.2f} coop {tco/tw:.2f}")
    return nb,tw,tr,tco
sim(0)
for cap in (32,48,64):
    sim(1,cap)
sim(1,64,2048,8)

def depth_rounds(lo,hi):
    op0=outStart[lo]
    idx=np.arange(lo,hi)
    mst=outStart[idx]+ll[idx]; sp=mst-off[idx]; m=ml[idx]
    coop=(m>18)|(off[idx]<m)
    far=(~coop)&(sp+m<=op0)
    n=hi-lo
    depth=np.zeros(n,dtype=int)
    # coop are serial barriers in the current scheme; here: treat coop as a sequence with depth like others but executed alone
    for j in range(n):
        if far[j]: depth[j]=0; continue
        d=0
        s0,s1=sp[j],sp[j]+m[j]
        if coop[j] and off[idx][j]<m[j]: s1=mst[j]  # overlapping: source is up to own start
        for k in range(j-1,-1,-1):
            if mst[k]+m[k]<=s0: 
                if outStart[idx][k]+outLen[idx][k] <= s0: break
            if far[k]: continue
            if mst[k]<s1 and mst[k]+m[k]>s0: d=max(d,depth[k])
        depth[j]=d+1
    return depth.max() if n else 0, int((~far & ~coop).sum())
def sim2(mode,cap=64,maxout=1024,maxwin=6):
    i=0;nb=0;td=0;tw=0;tn=0
    while i<len(seqs)-1:
        if not plain(i): i+=1;continue
        j=i;ip0=S[i,0];w=0
        if mode==0:
            while j<len(seqs)-1 and plain(j) and S[j,0]-ip0<64 and (S[j,0]+1+ll[j]-ip0)<=64 and outStart[j]+outLen[j]-outStart[i]<=1024: j+=1
            w=1
        else:
            wstart=ip0;w=1
            while j<len(seqs)-1 and plain(j) and (j-i)<cap and outStart[j]+outLen[j]-outStart[i]<=maxout:
                if S[j,0]-wstart>=64:
                    if w==maxwin: break
                    wstart=S[j,0]; w+=1
                j+=1
        if j==i: i+=1;continue
        d,nn=depth_rounds(i,j); nb+=1;td+=d;tw+=w;tn+=nn
        i=j
    print(f"mode {mode} cap {cap}: batches {nb} true depth/batch {td/nb:.2f} near seqs/batch {tn/nb:.1f} | per window {td/tw:.2f}")
sim2(0); sim2(1,32); sim2(1,64)
