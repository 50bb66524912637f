import re, sys
M32=0xffffffff; M64=(1<<64)-1
def run(lines, a_words, b_words):
    V={}; S={}; VCC=0
    loads=[]; out={}
    def sv(tok):
        tok=tok.strip()
        if tok.startswith('v['):
            lo,hi=map(int,re.match(r'v\[(\d+):(\d+)\]',tok).groups()); return ('v',lo,hi)
        if tok.startswith('s['):
            lo,hi=map(int,re.match(r's\[(\d+):(\d+)\]',tok).groups()); return ('s',lo,hi)
        if re.match(r'^v\d+$',tok): return ('v',int(tok[1:]),int(tok[1:]))
        if re.match(r'^s\d+$',tok): return ('s',int(tok[1:]),int(tok[1:]))
        if tok=='vcc': return ('vcc',0,0)
        import struct
        if re.match(r'^-?\d+\.\d+$',tok):
            return ('imm', struct.unpack('<I',struct.pack('<f',float(tok)))[0], 0)
        return ('imm', int(tok,0), 0)
    def rd(tok, width=32):
        k,lo,hi=sv(tok)
        if k=='imm':
            return lo & (M64 if width==64 else M32) if lo>=0 else (lo & (M64 if width==64 else M32))
        reg=V if k=='v' else S
        if k=='vcc': return VCC
        if width==64:
            if hi==lo:  # 32-bit reg used as 64? not expected
                return reg.get(lo,0)
            return reg.get(lo,0) | (reg.get(lo+1,0)<<32)
        return reg.get(lo,0)
    def wr(tok,val,width=32):
        nonlocal VCC
        k,lo,hi=sv(tok)
        if k=='vcc': VCC=val; return
        reg=V if k=='v' else S
        if width==64:
            reg[lo]=val&M32; reg[lo+1]=(val>>32)&M32
        else: reg[lo]=val&M32
    nload=0
    for ln in lines:
        ln=ln.split(';')[0].strip()
        if not ln or ln.endswith(':') or ln.startswith('.'): continue
        m=re.match(r'(\S+)\s*(.*)',ln); op=m.group(1); args=[x.strip() for x in re.split(r',\s*(?![^\[]*\])',m.group(2))] if m.group(2) else []
        if op in ('s_waitcnt','s_nop','s_endpgm','s_load_dword','s_load_dwordx4','s_load_dwordx2','s_and_b32','s_mul_i32','v_add_lshl_u32','v_lshlrev_b64') and op!='v_lshlrev_b64': continue
        if op=='global_load_dwordx4':
            k,lo,hi=sv(args[0]); src=[a_words[0:4],a_words[4:8],b_words[0:4],b_words[4:8]][nload]; nload+=1
            for i in range(4): V[lo+i]=src[i]
        elif op=='global_store_dwordx4':
            k,lo,hi=sv(args[1]); off=16 if 'offset:16' in ln else 0
            out[off]=[V.get(lo+i,0) for i in range(4)]
        elif op=='v_mov_b32_e32': wr(args[0], rd(args[1]))
        elif op in ('s_mov_b32',): wr(args[0], rd(args[1]))
        elif op=='s_movk_i32':
            v=int(args[1],0); v=v-0x10000 if v&0x8000 else v; wr(args[0], v&M32)
        elif op=='s_brev_b32':
            v=rd(args[1]); wr(args[0], int('{:032b}'.format(v)[::-1],2))
        elif op=='v_and_b32_e32': wr(args[0], rd(args[1])&rd(args[2]))
        elif op=='v_or_b32_e32': wr(args[0], rd(args[1])|rd(args[2]))
        elif op=='v_alignbit_b32':
            hi_,lo_,sh=rd(args[1]),rd(args[2]),rd(args[3])&31; wr(args[0], (((hi_<<32)|lo_)>>sh)&M32)
        elif op=='v_mul_lo_u32': wr(args[0], (rd(args[1])*rd(args[2]))&M32)
        elif op=='v_mul_u32_u24_e32': wr(args[0], ((rd(args[1])&0xffffff)*(rd(args[2])&0xffffff))&M32)
        elif op=='v_mul_hi_u32_u24_e32': wr(args[0], (((rd(args[1])&0xffffff)*(rd(args[2])&0xffffff))>>32)&M32)
        elif op=='v_lshrrev_b32_e32': wr(args[0], rd(args[2])>>(rd(args[1])&31))
        elif op=='v_lshlrev_b32_e32': wr(args[0], (rd(args[2])<<(rd(args[1])&31))&M32)
        elif op=='v_ashrrev_i32_e32':
            v=rd(args[2]); v=v-(1<<32) if v>>31 else v; wr(args[0], (v>>(rd(args[1])&31))&M32)
        elif op=='v_sub_u32_e32': wr(args[0], (rd(args[1])-rd(args[2]))&M32)
        elif op=='v_add_u32_e32': wr(args[0], (rd(args[1])+rd(args[2]))&M32)
        elif op=='v_mad_u64_u32':
            r=rd(args[2])*rd(args[3])+rd(args[4],64); wr(args[1], (r>>64)&1, 64 if sv(args[1])[0]!='vcc' else 32); wr(args[0], r&M64, 64)
        elif op=='v_lshl_add_u64':
            r=((rd(args[1],64)<<rd(args[2]))+ (rd(args[3],64) if sv(args[3])[0]!='imm' else (rd(args[3],64)))) & M64; wr(args[0], r, 64)
        elif op=='v_lshrrev_b64': wr(args[0], rd(args[2],64)>>(rd(args[1])&63), 64)
        elif op=='v_lshlrev_b64':
            # address computation or data? handle generally
            wr(args[0], (rd(args[2],64)<<(rd(args[1])&63))&M64, 64)
        elif op=='v_lshl_or_b32': wr(args[0], ((rd(args[1])<<(rd(args[2])&31))|rd(args[3]))&M32)
        elif op=='v_bfe_u32': wr(args[0], (rd(args[1])>>(rd(args[2])&31)) & ((1<<(rd(args[3])&31))-1))
        elif op=='v_bfi_b32':
            s0,s1,s2=rd(args[1]),rd(args[2]),rd(args[3]); wr(args[0], (s0&s1)|(~s0&s2&M32))
        elif op=='v_and_or_b32': wr(args[0], (rd(args[1])&rd(args[2]))|rd(args[3]))
        elif op=='v_cmp_eq_u64_e32': VCC = 1 if rd(args[1],64)==rd(args[2],64) else 0
        elif op=='v_cndmask_b32_e32':
            wr(args[0], rd(args[2]) if VCC else rd(args[1]))
        else:
            raise Exception('unhandled '+ln)
    return out
if __name__=='__main__':
    lines=open(sys.argv[1]).read().split('\n')
    p=8444461749428370424248824938781546531375899335154063827935233455917409239041
    def words(x): return [(x>>(32*i))&M32 for i in range(8)]
    for a,b in ((1,1),(2,3),(p-1,p-2),(0x1234567890abcdef<<100, 0xfedcba9876543210fedcba<<60)):
        o=run(lines, words(a), words(b))
        r=sum(w<<(32*i) for i,w in enumerate(o[0]+o[16]))
        e=a*b*pow(2,-256,p)%p
        print(hex(r)); print(hex(e), r==e)

def trace():
    lines=open('/tmp/bls_mul.s').read().split('\n')
    # re-run with tracing of lshrrev_b64 by 29
    import types
    src=open('/tmp/emu.py').read()
