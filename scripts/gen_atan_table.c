#include <quadmath.h>
#include <stdio.h>
static void split(__float128 v, double* hi, double* lo){ *hi=(double)v; *lo=(double)(v-(__float128)*hi); }
int main(){
  double h,l;
  printf("// atan(k/64), k = 0..64, as unevaluated sums hi + lo (generated with libquadmath atanq; scripts/gen_atan_table.c)\n");
  printf("MCP_ATAN_CONST double kAtanHi[65] = {\n");
  for(int k=0;k<=64;k++){ split(atanq((__float128)k/64), &h,&l); printf("  %a,%s", h, (k%4==3)?"\n":""); } printf("\n};\n");
  printf("MCP_ATAN_CONST double kAtanLo[65] = {\n");
  for(int k=0;k<=64;k++){ split(atanq((__float128)k/64), &h,&l); printf("  %a,%s", l, (k%4==3)?"\n":""); } printf("\n};\n");
  split(M_PI_2q,&h,&l); printf("// pi/2\n#define MCP_PIO2_HI %a\n#define MCP_PIO2_LO %a\n",h,l);
  __float128 one=1;
  split(one/3,&h,&l); printf("#define MCP_C3_HI %a\n#define MCP_C3_LO %a\n",h,l);
  split(one/5,&h,&l); printf("#define MCP_C5_HI %a\n#define MCP_C5_LO %a\n",h,l);
  split(one/7,&h,&l); printf("#define MCP_C7_HI %a\n#define MCP_C7_LO %a\n",h,l);
  return 0; }
