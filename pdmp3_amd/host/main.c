/* Same contract as the reference's main.c (main.c:1-6): every argument is a
 * file name, handed to pdmp3() without the program name. */
#include "../../include/pdmp3.h"
int main(int ac, char** av) {
  if (ac < 2) return 1;
  pdmp3(++av);
  return 0;
}
