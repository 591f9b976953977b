import itertools
GROUPS=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS+= [[l+32 for l in g] for g in GROUPS]
def conflicts(addr_of_lane):
    worst=0
    for g in GROUPS:
        slots={}
        for l in g:
            a=addr_of_lane(l)
            slots.setdefault((a//16)%16,set()).add(a)
        worst=max(worst,max(len(v) for v in slots.values()))
    return worst
def check(RB,COUT,pi,f):
    CPL=COUT//16*4; SPR=RB//16; NB=COUT//16
    w=0
    for nb in range(NB):
        def addr(l):
            lrow=l&15; lgrp=l>>4
            n=(lrow>>2)*CPL+nb*4+(lrow&3)
            s=lgrp%SPR
            return pi(n)*RB+((s^f(n))%SPR)*16
        w=max(w,conflicts(addr))
    return w
# candidates
ident=lambda n:n
print('RB64 C32 plain', check(64,32,ident,lambda n:0))
g=(0,2,3,1)
print('RB64 C32 g', check(64,32,ident,lambda n:g[(n>>3)&3]))
print('RB32 C16 plain', check(32,16,ident,lambda n:0))
print('RB32 C32 plain', check(32,32,ident,lambda n:0))
pi32=lambda n:(n&3)|((n>>3)<<2)|(((n>>2)&1)<<4)
print('RB32 C32 pi', check(32,32,pi32,lambda n:0))
print('RB64 C64 plain', check(64,64,ident,lambda n:0))
# search f over tables of (n>>k) for RB64 C64 (CPL=16): n=(lrow>>2)*16+nb*4+(lrow&3)
for perm in itertools.product(range(4),repeat=4):
    if check(64,64,ident,lambda n:perm[(n>>4)&3])==1: print('RB64 C64 f by n>>4', perm); break
for bit in range(5):
    print('RB32 C32 f=bit',bit, check(32,32,ident,lambda n:(n>>bit)&1))
for b1 in range(5):
  for b2 in range(5):
    if check(32,32,ident,lambda n:((n>>b1)^(n>>b2))&1)==1: print('RB32 C32 xor bits',b1,b2)
