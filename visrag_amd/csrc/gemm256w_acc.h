// Hand-allocated accumulation registers of the one-wave-per-SIMD 256 x 256 tile (gemm256w.hip, search256w.hip).
// Files that include this are disassembled by build.py: outside their own asm there must be no
// accumulation-register traffic at all.
#pragma once

// The 256 accumulators of a lane are HAND-ALLOCATED: accumulator n = strip * 8 + fragment (strip = 16-row
// strip of the wave's 128 rows, fragment = 16-column fragment of its 128 columns) lives in a[4n : 4n+3].
// hipcc cannot keep 64 four-register accumulators in a file they fill exactly: as builtin values, or as
// "+a" asm operands, it rotates them through other registers around every MFMA (hundreds of v_accvgpr
// moves and scratch spills per K-step), and with physical-register constraints it copies them in and out
// of VGPRs around every statement.  So the MFMAs name their accumulation registers in the asm text, the
// registers are zeroed and read back by asm as well, and the compiler only sees them as clobbers (which
// also makes them count in the kernel's register allocation).  Its hazard bookkeeping does not see into
// the asm: dependent MFMAs are 64 instructions apart, and explicit wait states separate the K-loop from
// the read-back.  (The compiler never uses accumulation registers on its own here: no builtin MFMAs, and
// the arch VGPRs do not spill.)
#define W_FOR_EACH_ACC(X) \
    X(0, "a[0:3]", "a0", "a1", "a2", "a3") \
    X(1, "a[4:7]", "a4", "a5", "a6", "a7") \
    X(2, "a[8:11]", "a8", "a9", "a10", "a11") \
    X(3, "a[12:15]", "a12", "a13", "a14", "a15") \
    X(4, "a[16:19]", "a16", "a17", "a18", "a19") \
    X(5, "a[20:23]", "a20", "a21", "a22", "a23") \
    X(6, "a[24:27]", "a24", "a25", "a26", "a27") \
    X(7, "a[28:31]", "a28", "a29", "a30", "a31") \
    X(8, "a[32:35]", "a32", "a33", "a34", "a35") \
    X(9, "a[36:39]", "a36", "a37", "a38", "a39") \
    X(10, "a[40:43]", "a40", "a41", "a42", "a43") \
    X(11, "a[44:47]", "a44", "a45", "a46", "a47") \
    X(12, "a[48:51]", "a48", "a49", "a50", "a51") \
    X(13, "a[52:55]", "a52", "a53", "a54", "a55") \
    X(14, "a[56:59]", "a56", "a57", "a58", "a59") \
    X(15, "a[60:63]", "a60", "a61", "a62", "a63") \
    X(16, "a[64:67]", "a64", "a65", "a66", "a67") \
    X(17, "a[68:71]", "a68", "a69", "a70", "a71") \
    X(18, "a[72:75]", "a72", "a73", "a74", "a75") \
    X(19, "a[76:79]", "a76", "a77", "a78", "a79") \
    X(20, "a[80:83]", "a80", "a81", "a82", "a83") \
    X(21, "a[84:87]", "a84", "a85", "a86", "a87") \
    X(22, "a[88:91]", "a88", "a89", "a90", "a91") \
    X(23, "a[92:95]", "a92", "a93", "a94", "a95") \
    X(24, "a[96:99]", "a96", "a97", "a98", "a99") \
    X(25, "a[100:103]", "a100", "a101", "a102", "a103") \
    X(26, "a[104:107]", "a104", "a105", "a106", "a107") \
    X(27, "a[108:111]", "a108", "a109", "a110", "a111") \
    X(28, "a[112:115]", "a112", "a113", "a114", "a115") \
    X(29, "a[116:119]", "a116", "a117", "a118", "a119") \
    X(30, "a[120:123]", "a120", "a121", "a122", "a123") \
    X(31, "a[124:127]", "a124", "a125", "a126", "a127") \
    X(32, "a[128:131]", "a128", "a129", "a130", "a131") \
    X(33, "a[132:135]", "a132", "a133", "a134", "a135") \
    X(34, "a[136:139]", "a136", "a137", "a138", "a139") \
    X(35, "a[140:143]", "a140", "a141", "a142", "a143") \
    X(36, "a[144:147]", "a144", "a145", "a146", "a147") \
    X(37, "a[148:151]", "a148", "a149", "a150", "a151") \
    X(38, "a[152:155]", "a152", "a153", "a154", "a155") \
    X(39, "a[156:159]", "a156", "a157", "a158", "a159") \
    X(40, "a[160:163]", "a160", "a161", "a162", "a163") \
    X(41, "a[164:167]", "a164", "a165", "a166", "a167") \
    X(42, "a[168:171]", "a168", "a169", "a170", "a171") \
    X(43, "a[172:175]", "a172", "a173", "a174", "a175") \
    X(44, "a[176:179]", "a176", "a177", "a178", "a179") \
    X(45, "a[180:183]", "a180", "a181", "a182", "a183") \
    X(46, "a[184:187]", "a184", "a185", "a186", "a187") \
    X(47, "a[188:191]", "a188", "a189", "a190", "a191") \
    X(48, "a[192:195]", "a192", "a193", "a194", "a195") \
    X(49, "a[196:199]", "a196", "a197", "a198", "a199") \
    X(50, "a[200:203]", "a200", "a201", "a202", "a203") \
    X(51, "a[204:207]", "a204", "a205", "a206", "a207") \
    X(52, "a[208:211]", "a208", "a209", "a210", "a211") \
    X(53, "a[212:215]", "a212", "a213", "a214", "a215") \
    X(54, "a[216:219]", "a216", "a217", "a218", "a219") \
    X(55, "a[220:223]", "a220", "a221", "a222", "a223") \
    X(56, "a[224:227]", "a224", "a225", "a226", "a227") \
    X(57, "a[228:231]", "a228", "a229", "a230", "a231") \
    X(58, "a[232:235]", "a232", "a233", "a234", "a235") \
    X(59, "a[236:239]", "a236", "a237", "a238", "a239") \
    X(60, "a[240:243]", "a240", "a241", "a242", "a243") \
    X(61, "a[244:247]", "a244", "a245", "a246", "a247") \
    X(62, "a[248:251]", "a248", "a249", "a250", "a251") \
    X(63, "a[252:255]", "a252", "a253", "a254", "a255")
#define W_MFMA(R, C0, C1, C2, C3, WV, AV) \
    asm volatile("v_mfma_f32_16x16x32_bf16 " R ", %0, %1, " R : : "v"(WV), "v"(AV) : C0, C1, C2, C3)
#define W_ZERO(n, R, C0, C1, C2, C3) \
    asm volatile("v_accvgpr_write_b32 " C0 ", 0\n\tv_accvgpr_write_b32 " C1 ", 0\n\tv_accvgpr_write_b32 " C2 \
                 ", 0\n\tv_accvgpr_write_b32 " C3 ", 0" : : : C0, C1, C2, C3);
#define W_READ(V, C0, C1, C2, C3) \
    asm volatile("v_accvgpr_read_b32 %0, " C0 "\n\tv_accvgpr_read_b32 %1, " C1 "\n\tv_accvgpr_read_b32 %2, " C2 \
                 "\n\tv_accvgpr_read_b32 %3, " C3 : "=v"(V[0]), "=v"(V[1]), "=v"(V[2]), "=v"(V[3]))

