#!/usr/bin/env python3
"""CPU experiment (numpy / scipy, float32 vs float64): where does the element-wise error of a float32 real FFT of pre-emphasised noise sit, and what
does the packed form (complex FFT of half the size + split step, the form every fft*c kernel uses) add to it?  Low bins after pre-emphasis carry
~1e-3 of the mid-band power, a few in a million 1e-7: there BOTH float32 pipelines are at their rounding floor.  Output (this container): the
packed form carries ~1.4 x the rms / ~2 x the max relative error of the direct float32 rfft on those bins, and it comes from the complex FFT, not
from the split step (an exact Z rounded to float32 + a float32 split is 3-4 x better than either).  DESIGN.md section 2."""
import numpy as np, scipy.fft as sf
rs=np.random.RandomState(0)
F=4000
x=(rs.uniform(-0.5,0.5,size=(F,401))).astype(np.float32)
# preemph + povey window, float64 truth
def prep(x, dt):
    x=x.astype(dt); y=x[:,1:]-dt(0.97)*x[:,:-1]
    n=np.arange(400); w=(0.5-0.5*np.cos(2*np.pi*n/399))**0.85
    y=y*w.astype(dt); out=np.zeros((x.shape[0],512),dt); out[:,:400]=y; return out
y64=prep(x,np.float64); y32=prep(x,np.float32)
P64=np.abs(np.fft.rfft(y64))**2
def relerr(P): return np.abs(P/P64-1)
# (a) direct float32 rfft (pocketfft in float32)
Pa=np.abs(sf.rfft(y32).astype(np.complex64))**2
# (b) packed: complex FFT256 in float32 (pocketfft c2c float32) + split in float64
z=(y32[:,0::2]+1j*y32[:,1::2]).astype(np.complex64)
Z32=sf.fft(z).astype(np.complex64)
def split(Z, dt):
    Z=Z.astype(np.complex128 if dt==np.float64 else np.complex64)
    k=np.arange(257); Zk=np.concatenate([Z,Z[:,:1]],1); Zp=np.conj(Zk[:,::-1])
    W=np.exp(-2j*np.pi*k/512).astype(Z.dtype)
    E=(Zk+Zp)*Z.dtype.type(0.5); O=(Zk-Zp)*Z.dtype.type(-0.5j)
    return E+W*O
Pb=np.abs(split(Z32,np.float64))**2
# (c) exact Z rounded to float32, split in float32
Z64=np.fft.fft(y64[:,0::2]+1j*y64[:,1::2])
Pc=np.abs(split(Z64.astype(np.complex64),np.float32).astype(np.complex128))**2
# (d) both float32
Pd=np.abs(split(Z32,np.float32).astype(np.complex128))**2
for name,P in (("direct f32 rfft",Pa),("f32 FFT256 + f64 split",Pb),("exact Z(f32-rounded) + f32 split",Pc),("f32 FFT256 + f32 split",Pd)):
    e=relerr(P)[:,1:9]; small=P64[:,1:9]<1e-4*np.median(P64,axis=1,keepdims=True)
    print(f"{name:36s} low bins 1-8: rms rel {np.sqrt((e**2).mean()):.2e} max {e.max():.2e}; tiny bins (n={small.sum()}): rms {np.sqrt((e[small]**2).mean()):.2e} max {e[small].max():.2e}")
