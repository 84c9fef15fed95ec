"""LDS bank-conflict model of the solver kernel's gather tables: replicates make_tables() of csrc/frp_ipm_lds.hip and counts, with the
per-instruction bank rules of MI355X_MICROARCH.md (ds_read_b64: two 32-lane groups, bank = double index mod 32; ds_write_b64: four 16-lane groups, mod 16),
the LDS cycles of every gather / store of a factorisation stage and of the vector sweeps.   python tools/lds_bank_sim.py"""
import itertools, sys
R_LIN=0;R_D=51;R_T=64;R_PHID=128;R_PHIPOS=145;R_PHI=154;R_HD=171;R_P=128;R_PV=219;R_PD=232;R_PHIB=245;R_CB=262;R_BC=21
R_PHIC=R_PHIB+R_BC;R_CC=R_CB+R_BC;R_HC=286;R_ZERO=287;R_ONE=288;R_DT=289;R_DUMP=290;R_DZ=291;R_ZERO2=R_ZERO+R_BC;RS=309
def hd_zero(i,j): return 3<=i<=6 and 3<=j<=6
def hd_pack(i,j):
    if i>j: i,j=j,i
    if hd_zero(i,j): return -1
    n=0
    for a in range(10):
        for b in range(a,10):
            if a==i and b==j: return n
            if not hd_zero(a,b): n+=1
    return -1
def zi_of(a): return a if a<4 else a+4
def hidx_of(a): return a if a<4 else (a-3 if 7<=a<=12 else -1)
def m_src(row,col):
    if row>12 or col>13: return R_ZERO
    if col==13: return R_D+row
    if row<4: return R_ONE if col==row else R_ZERO
    i=row-4;bi=i//3;ii=i%3
    if col<4:
        if col==3: return R_LIN+36+ii if bi==0 else (R_LIN+39+ii if bi==1 else R_ZERO)
        if bi==1: return R_LIN+42+ii*3+col
        if bi==2: return R_DT if ii==col else R_ZERO
        return R_ZERO
    j=col-4;bj=j//3;jj=j%3
    if bi==0: return (R_ONE if ii==jj else R_ZERO) if bj==0 else R_LIN+(0 if bj==1 else 9)+ii*3+jj
    if bi==1: return R_ZERO if bj==0 else R_LIN+(18 if bj==1 else 27)+ii*3+jj
    return R_ONE if (bj==2 and ii==jj) else R_ZERO
def c_src(row,col,which):
    o1=o2=o3=R_ZERO
    if row<=12:
        if col==13:
            o1=R_PHI+zi_of(row)
            if 4<=row<=6: o2=R_CC+(row-4)
        elif col<=12:
            if col==row: o1=R_PHID+zi_of(row)
            if 4<=row<=6 and 4<=col<=6: o2=R_PHIPOS+(row-4)*3+(col-4)
            hr=hidx_of(row);hc=hidx_of(col)
            if hr>=0 and hc>=0 and hd_pack(hr,hc)>=0: o3=R_HD+hd_pack(hr,hc)
    return (o1,o2,o3)[which]
def tables():
    t={}
    for lane in range(64):
        g=lane>>4;c=lane&15;qk=lane>>4;qI=(lane>>2)&3;qj=lane&3
        for r in range(4):
            trow=4*r+g
            t.setdefault(('M',r),[]).append(m_src(trow,c))
            t.setdefault(('C1',r),[]).append(c_src(trow,c,0))
            t.setdefault(('C2',r),[]).append(c_src(trow,c,1))
            t.setdefault(('C3',r),[]).append(c_src(trow,c,2))
            t.setdefault(('PP',r),[]).append(R_P+trow*(trow+1)//2+c if (trow<=12 and c<=trow) else R_DUMP)
            t.setdefault(('PD',r),[]).append(R_PD+trow if (c==13 and trow<=12) else R_DUMP)
            row=4*qI+qj;col=4*((qI+r)&3)+qk
            t.setdefault(('MT',r),[]).append(R_ZERO if col<4 else (R_ONE if (row==13 and col==13) else m_src(row,col)))
            t.setdefault(('MTT',r),[]).append(R_ZERO if (row<4 or row>12) else m_src(col,row))
            hi=max(row,col);lo=min(row,col)
            t.setdefault(('P4',r),[]).append(R_P+hi*(hi+1)//2+lo if (row<=12 and col<=12) else R_ZERO)
        t.setdefault(('MU',0),[]).append(m_src(4*qI+qj,qk))
        t.setdefault(('MTTU',0),[]).append(m_src(4*qI+qk,qj))
        t.setdefault(('TS',0),[]).append(R_T+16*qj+4*qI+qk)
        t.setdefault(('T',0),[]).append(R_T+lane)
    return t
def read_cycles(addrs, nb=32, groups=((0,32),(32,64))):
    cyc=0
    for a,b in groups:
        per={}
        for x in addrs[a:b]:
            per.setdefault(x%nb,set()).add(x)
        cyc+=max(len(s) for s in per.values())
    return cyc
def write_cycles(addrs):
    # ds_write_b64: 4 x 16 contiguous, bank (a/4)%32 -> doubles mod 16
    return read_cycles(addrs, nb=16, groups=((0,16),(16,32),(32,48),(48,64)))
if __name__=="__main__":
    t=tables()
    tot=0;base=0
    for k,v in t.items():
        if k[0] in('PP','PD'): c=write_cycles(v); b=4
        else: c=read_cycles(v); b=2
        print(k,c,'(min %d)'%b)
